"""Co-residency hunt, round 4: the round-3 reproducer (side_race.py) with a selectable library build and a full dump of
the first diverging GroupNorm-backward workgroups (every input of the kernel, the result and the clean-run result), so the
wrong values can be explained off-line.

  PROBE_LIB=<path of a libstk build>  PROBE_OUT=<dir>  PROBE_W1=0|1  PROBE_DELAYS=200000,700000,...  python tools/_probe/side_race2.py
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  sys.path.insert(0, p)
import torch
from importlib import import_module
import soft_truncation_amd as st
from _model_util import build_pair, tiny_config
from _model_cases import _inputs
G = import_module('soft-truncation_amd.engine.graph')
E = import_module('soft-truncation_amd.engine.executor')
LIBMOD = import_module('soft-truncation_amd.engine.lib')
if os.environ.get('PROBE_LIB'):
  LIBMOD.PRODUCT_LIB = os.path.abspath(os.environ['PROBE_LIB'])
OUT = os.environ.get('PROBE_OUT', os.path.join(ROOT, 'gpurun_out', 'race'))
TAG = os.environ.get('PROBE_TAG', 'run')
os.makedirs(OUT, exist_ok=True)
lib = st.engine.lib.load()
print('library:', lib.path, flush=True)
cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, 'wide'), lib)
dev = cfg.device
x, t, cond = _inputs(cfg, sde, 96)
go = torch.randn(96, 3, 16, 16, generator=torch.Generator().manual_seed(5)).to(dev)
W1 = os.environ.get('PROBE_W1', '0') == '1'
SHORT = os.environ.get('PROBE_SHORTCUT', '0') == '1'


def run(side=True):
  G._SIDE_SHORTCUT = SHORT and side; G._SIDE_WGRAD1 = W1 and side
  model.module.engine().use_side = side
  model.zero_grad()
  xg = x.clone().to(dev).requires_grad_(True)
  y = model(xg, cond.to(dev))
  (y * go).sum().backward()
  torch.cuda.synchronize()
  return [p.grad.detach().clone() for p in model.parameters()], xg.grad.clone()


model.eval()
run(side=False)
CUR = [None]
orig_rb = E.Executor._run_backward
def rb(self, c, gout, param_grads):
  CUR[0] = c
  return orig_rb(self, c, gout, param_grads)
E.Executor._run_backward = rb
ex = model.module.engine()
prog = next(iter(ex.programs.values()))
REC = {}          # op name -> dict of tensors of the current run
WATCH = None      # names of the GroupNorm ops to record (None = all with by-products)


def view(c, t, grad):
  if t is None:
    return None
  if t.space == 'param':
    buf = ex.flat.grad if grad else ex.flat.data
    return None
  if grad:
    if t.goff is None:
      return torch.zeros(1, device=dev)
    return c.gact[t.goff:t.goff + t.numel].view(t.shape)
  return c.act[t.off:t.off + t.numel].view(t.shape)


gn_ops = [op for op in prog.graph.ops if type(op).__name__ == 'GroupNormAct' and (op.dy_cons is not None or op.add_from is not None)]
print('GroupNorm ops with by-products:', [(op.y.name, op.C1, op.C2, op.HW) for op in gn_ops], flush=True)
for op in gn_ops:
  def wrap(op):
    ob = op.backward
    def b(rt):
      c = CUR[0]
      if WATCH is not None and op.y.name not in WATCH:
        return ob(rt)
      r = {}
      r['pre_dx1'] = view(c, op.x1, True).clone()
      if op.add_from is not None:
        r['add'] = view(c, op.add_from.y, True).clone()
        r['add_scale'] = 1.0 / op.add_from.out_div
      r['dy'] = view(c, op.y, True).clone()
      r['x1'] = view(c, op.x1, False).clone()
      r['mean'] = view(c, op.mean, False).clone()
      r['rstd'] = view(c, op.rstd, False).clone()
      ob(rt)
      r['dx1'] = view(c, op.x1, True).clone()
      r['beta1'] = op.b(op.x1)
      REC[op.y.name] = r
    op.backward = b
  wrap(op)


def run_chk(delay_name, delay):
  REC.clear()
  G._SIDE_DELAY = delay if delay_name else 0
  G._SIDE_DELAY_FILTER = (lambda n: n == delay_name) if delay_name else None
  g, gx = run(True)
  return g, {k: v for k, v in REC.items()}


base_g, base_rec = run_chk(None, 0)
g2, rec2 = run_chk(None, 0)
print('clean two-stream runs identical:', all(torch.equal(a, b) for a, b in zip(base_g, g2)), flush=True)
seen = []
G._SIDE_DELAY = 1
G._SIDE_DELAY_FILTER = lambda n: (seen.append(n), False)[1]
run(True)
sites = []
for n in seen:
  if n not in sites:
    sites.append(n)
sites = sites[8:]
print('delay sites:', sites, flush=True)
delays = [int(v) for v in os.environ.get('PROBE_DELAYS', '200000,400000,700000,1000000,1500000,2200000,3000000').split(',')]
reps = int(os.environ.get('PROBE_REPS', '2'))
found = 0
runs = 0
dumped = 0
summary = []
for delay in delays:
  for nm in sites:
    for rep in range(reps):
      runs += 1
      g, rec = run_chk(nm, delay)
      nbad = sum(1 for a, b in zip(g, base_g) if not torch.equal(a, b))
      if not nbad:
        continue
      found += 1
      # the first GroupNorm (in backward order = recording order) whose result differs although its inputs agree
      for name in rec:
        r, b = rec[name], base_rec[name]
        same_in = all(torch.equal(r[k], b[k]) for k in ('pre_dx1', 'dy', 'x1', 'mean', 'rstd') + (('add',) if 'add' in r else ()))
        bad = r['dx1'] != b['dx1']
        if bad.any():
          idx = bad.nonzero()
          C = r['dx1'].shape[1]
          opx = next(o for o in gn_ops if o.y.name == name)
          cpg = (opx.C1 + opx.C2) // opx.G
          groups = sorted(set((int(i[0]), int(i[1]) // cpg) for i in idx))
          flatpos = bad[:, :bad.shape[1] // cpg * cpg].reshape(bad.shape[0], bad.shape[1] // cpg, -1).nonzero()
          lanes = sorted(set(int(p[2]) // 4 for p in flatpos))
          comps = sorted(set(int(p[2]) % 4 for p in flatpos))
          d = (r['dx1'] - b['dx1'])[bad]
          line = dict(delay=delay, site=nm, op=name, inputs_equal=bool(same_in), n_bad=int(bad.sum()), groups=len(groups), lanes=lanes[:70],
                      comps=comps, max_abs=float(d.abs().max()), max_rel=float((d.abs() / b['dx1'][bad].abs().clamp_min(1e-30)).max()))
          summary.append(line)
          print(json.dumps(line), flush=True)
          if same_in and dumped < 6:
            dumped += 1
            n0, g0 = groups[0]
            sl = (slice(n0, n0 + 1), slice(cpg * g0, cpg * g0 + cpg))
            blob = {k: (v[sl].cpu() if torch.is_tensor(v) and v.dim() == 4 else (v.cpu() if torch.is_tensor(v) else v)) for k, v in r.items()}
            blob['ref_dx1'] = b['dx1'][sl].cpu()
            blob['n'] = n0; blob['g'] = g0; blob['op'] = name; blob['groups'] = groups
            blob['beta'] = ex.flat.data[opx.beta_t.off:opx.beta_t.off + opx.beta_t.numel].cpu()
            blob['cpg'] = cpg
            blob['gamma'] = ex.flat.data[opx.gamma.off:opx.gamma.off + opx.gamma.numel].cpu() if hasattr(ex, 'flat') else None
            torch.save(blob, os.path.join(OUT, f'{TAG}_dump{dumped}.pt'))
          break
print(json.dumps(dict(tag=TAG, lib=lib.path, runs=runs, diverged=found)), flush=True)
with open(os.path.join(OUT, f'{TAG}_summary.json'), 'w') as f:
  json.dump(dict(tag=TAG, lib=lib.path, runs=runs, diverged=found, cases=summary), f, indent=1)
