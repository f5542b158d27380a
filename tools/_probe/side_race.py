"""Debug: which parameter gradients differ between runs with the shortcut backward on the side stream and on the main one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  sys.path.insert(0, p)
import torch
from importlib import import_module
import soft_truncation_amd as st
from _model_util import build_pair, tiny_config
from _model_cases import _inputs
G = import_module('soft-truncation_amd.engine.graph')
lib = st.engine.lib.load()
cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, 'wide'), lib)
dev = cfg.device
x, t, cond = _inputs(cfg, sde, 96)
go = torch.randn(96, 3, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
names = [n for n, p in model.named_parameters()]
POISON = [None]
def run(shortcut, wgrad1=True, side=True):
  G._SIDE_SHORTCUT = shortcut; G._SIDE_WGRAD1 = wgrad1
  model.module.engine().use_side = side
  model.zero_grad()
  xg = x.clone().to(dev).requires_grad_(True)
  y = model(xg, cond.to(dev))
  if POISON[0] is not None:
    for prog in model.module.engine().programs.values():
      prog.ws.fill_(POISON[0])
      if prog.ws2 is not None: prog.ws2.fill_(POISON[0])
  (y * go).sum().backward()
  torch.cuda.synchronize()
  return [p.grad.detach().clone() for p in model.parameters()], xg.grad.clone()
model.eval()
base, bx = run(False, False, side=False)
E = import_module('soft-truncation_amd.engine.executor')
CUR = [None]
CHK = []
SAVE = {}
PRE = [None]
orig_rb = E.Executor._run_backward
def rb(self, c, gout, param_grads):
  CUR[0] = c
  return orig_rb(self, c, gout, param_grads)
E.Executor._run_backward = rb
ex = model.module.engine()
prog = next(iter(ex.programs.values()))
for op in prog.graph.ops:
  def wrap(op):
    ob = op.backward
    def b(rt):
      c = CUR[0]
      if type(op).__name__ == 'GroupNormAct' and op.y.name == 'd1.0.gn0':
        t = op.x1
        PRE[0] = c.gact[t.goff:t.goff + t.numel].clone().view(t.shape)
      ob(rt)
      if type(op).__name__ == 'GroupNormAct' and op.y.name in ('d0.down.gn1', 'd1.0.gn0'):
        t = op.x1
        SAVE[op.y.name] = (c.gact[t.goff:t.goff + t.numel].clone().view(t.shape), c.gact[op.y.goff:op.y.goff + op.y.numel].clone().view(op.y.shape),
                           c.act[t.off:t.off + t.numel].clone().view(t.shape))
      for t in op.inputs:
        if t is not None and t.needs_grad and t.space == 'act' and t.goff is not None:
          CHK.append((type(op).__name__ + ':' + getattr(op.y, 'name', '?') + '->d(' + t.name + ')', c.gact[t.goff:t.goff + t.numel].double().abs().sum()))
    op.backward = b
  wrap(op)
W1 = [os.environ.get('PROBE_W1', '0') == '1']
def run_chk(delay_name, delay=1000000):
  del CHK[:]
  G._SIDE_DELAY = delay if delay_name else 0
  G._SIDE_DELAY_FILTER = (lambda n: n == delay_name) if delay_name else None
  g, gx = run(False, W1[0], True)
  return [(n, float(v)) for n, v in CHK], sum(1 for a, b in zip(g, base) if not torch.equal(a, b))
ref, nb = run_chk(None)
REF_SAVE = {k: tuple(t.clone() for t in v) for k, v in SAVE.items()}
REF_PRE = PRE[0].clone()
base = [p.grad.detach().clone() for p in model.parameters()]
print('clean run: differing tensors', nb, 'checksums', len(ref))
seen2 = []
G._SIDE_DELAY = 1
G._SIDE_DELAY_FILTER = lambda n: (seen2.append(n), False)[1]
run(False, W1[0], True)
order_names = []
for n in seen2:
  if n not in order_names: order_names.append(n)
order_names = order_names[8:]
print('delay sites:', order_names)
found = 0
for delay in (200000, 400000, 700000, 1000000, 1500000, 2200000, 3000000):
  for nm in order_names:
    for rep in range(2):
      got, nb = run_chk(nm, delay)
      first = next((i for i, (a, b) in enumerate(zip(ref, got)) if a[1] != b[1]), None)
      if first is not None:
        found += 1
        print(f'delay {delay} at {nm}: differing tensors {nb}; first diverging checksum: {first} {got[first][0] if first is not None else ""}')
        if first is not None and found <= 4:
          for k in SAVE:
            dx, dy, xx = SAVE[k]; rdx, rdy, rxx = REF_SAVE[k]
            bad = (dx != rdx)
            if bad.any():
              idx = bad.nonzero()
              ng = sorted(set((int(i[0]), int(i[1]) // 4) for i in idx))
              print(f'   {k}: dx differs in {int(bad.sum())} of {bad.numel()} elements; (sample, group) pairs touched: {len(ng)} first {ng[:6]}; dy equal {torch.equal(dy, rdy)}; x equal {torch.equal(xx, rxx)}')
              if k == 'd1.0.gn0':
                pb = (PRE[0] != REF_PRE)
                print(f'      dx BEFORE the op (written by earlier kernels) differs in {int(pb.sum())} elements; same positions as after: {bool((pb == bad).all())}')
                flat = bad.reshape(bad.shape[0], -1)
                pos = flat.nonzero()
                print('      flat positions (sample, offset in sample):', [(int(a), int(b)) for a, b in pos[:20]])
              n0, g0 = ng[0]
              blk = dx[n0, 4 * g0:4 * g0 + 4].reshape(-1); rblk = rdx[n0, 4 * g0:4 * g0 + 4].reshape(-1)
              nb_ = int((blk != rblk).sum())
              print(f'      group ({n0},{g0}): {nb_} of {blk.numel()} elements differ; got[:6] {blk[:6].tolist()} ref[:6] {rblk[:6].tolist()}; ratio/diff stats: max|d| {float((blk - rblk).abs().max()):.3e}')
        if first is not None and found <= 4:
          for i in range(max(0, first - 2), min(len(got), first + 3)):
            print('      ', i, got[i][0], ref[i][1], got[i][1])
print('runs with a divergence:', found)
