"""Worker for tests/test_ddp_gloo.py: one rank of a world_size-2 gloo job on CPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
  if p not in sys.path:
    sys.path.insert(0, p)

import contextlib

import numpy as np
import torch
import torch.distributed as dist


class ShardedDraws:
  """Rank r of W gets rows [r*n, (r+1)*n) of the noise a single process would draw for the W*n batch."""

  def __init__(self, seed, rank, world):
    self.g = torch.Generator().manual_seed(seed)
    self.rank, self.world = rank, world

  def _cut(self, full, n):
    return full[self.rank * n:(self.rank + 1) * n].clone()

  def rand(self, *size, device=None, **kw):
    n = size[0]
    return self._cut(torch.empty(n * self.world, *size[1:]).uniform_(generator=self.g), n).to(device or 'cpu')

  def randn_like(self, x, **kw):
    n = x.shape[0]
    return self._cut(torch.empty(n * self.world, *x.shape[1:]).normal_(generator=self.g), n).to(x.device)


@contextlib.contextmanager
def sharded_rng(seed, rank, world):
  d = ShardedDraws(seed, rank, world)
  saved = (torch.rand, torch.randn_like)
  torch.rand, torch.randn_like = d.rand, d.randn_like
  try:
    yield
  finally:
    torch.rand, torch.randn_like = saved


def build(st, lib, family='vp', mixed=False, device='cpu'):
  from _model_util import randomize_, tiny_config
  cfg = tiny_config(st, family, device=device)
  cfg.optim.warmup = 2
  cfg.training.mixed = bool(mixed)      # two network evaluations (and two engine backwards) per loss: losses.py:295-320
  sde = st.sde_lib.get_sde(cfg, None)
  net = st.models.ncsnpp.NCSNpp(cfg, sde)
  net.set_backend(lib)
  net = net.to(cfg.device)
  randomize_(net, 0)
  model = st.models.utils.DataParallel(net)
  net.engine().ensure_flat()
  opt = st.losses.get_optimizer(cfg, model.parameters())
  opt._backend = lib
  ema = st.models.ema.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
  ema.set_backend(lib)
  state = dict(optimizer=opt, model=model, ema=ema, step=0)
  step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  return cfg, state, step_fn


def run_steps(st, lib, rank, world, steps, global_batch, mixed=False, family='vp', device='cpu'):
  cfg, state, step_fn = build(st, lib, family=family, mixed=mixed, device=device)
  st.engine.ddp.seed_everything(123)          # numpy shared -> the same t_min on every rank
  losses = []
  for i in range(steps):
    full = st.datasets.synthetic_batch(cfg, global_batch, generator=torch.Generator().manual_seed(100 + i))
    local = st.engine.ddp.shard_batch(full) if world > 1 else full
    with sharded_rng(50 + i, rank, world):
      losses.append(step_fn(state, local.to(cfg.device)))
  ex = state['model'].module.engine()
  flat = ex.flat
  run_steps.graph_replays = ex.graph_replays
  return torch.cat(losses), flat.data[:flat.n_train].detach().cpu().clone(), state['ema']._shadow.detach().cpu().clone()


def worker(rank, world, port, outdir):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.set_num_threads(2)
  import soft_truncation_amd as st
  lib = st.engine.lib.load_path(os.path.join(ROOT, 'oracle', 'libstk_ref.so'))
  # 1) the bucketed all-reduce itself
  buf = torch.arange(1000, dtype=torch.float32) * (rank + 1)
  nb = st.engine.ddp.allreduce_flat_(buf, 1000, bucket_mb=0.001, average=True)     # ~262-element buckets
  assert nb == 4
  assert torch.allclose(buf, torch.arange(1000, dtype=torch.float32) * 1.5)
  # 2) two training steps on this rank's shard, gradients exchanged DURING the last backward (small buckets so that the
  #    tiny model has several), then the same with the exchange after the backward: bit-identical
  os.environ['STK_DDP_BUCKET_MB'] = '0.05'
  calls = []
  real = dist.all_reduce
  dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real(*a, **k))[1]
  assert st.losses.OVERLAP_EXCHANGE
  losses, params, shadow = run_steps(st, lib, rank, world, steps=2, global_batch=4)
  n_overlapped = len(calls)
  st.losses.OVERLAP_EXCHANGE = False
  del calls[:]
  losses_b, params_b, shadow_b = run_steps(st, lib, rank, world, steps=2, global_batch=4)
  dist.all_reduce = real
  assert n_overlapped >= 2 * 3, n_overlapped            # >= 3 buckets per step went out from inside the backward
  assert torch.equal(params, params_b) and torch.equal(shadow, shadow_b) and torch.equal(losses, losses_b)
  # 3) training.mixed (the reference's ddpmpp_*_st_deepest configs): every loss evaluates the network twice, so ONE
  #    .backward() runs two engine backwards into the same flat buffer.  Buckets may only leave with the last of them:
  #    overlapped == plain, bit for bit, and buckets still went out from inside the backward.
  st.losses.OVERLAP_EXCHANGE = True
  dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real(*a, **k))[1]
  del calls[:]
  m_losses, m_params, m_shadow = run_steps(st, lib, rank, world, steps=2, global_batch=8, mixed=True)
  n_mixed = len(calls)
  st.losses.OVERLAP_EXCHANGE = False
  m_losses_b, m_params_b, m_shadow_b = run_steps(st, lib, rank, world, steps=2, global_batch=8, mixed=True)
  dist.all_reduce = real
  st.losses.OVERLAP_EXCHANGE = True
  assert n_mixed >= 2 * 3, n_mixed
  assert torch.equal(m_params, m_params_b) and torch.equal(m_shadow, m_shadow_b) and torch.equal(m_losses, m_losses_b)
  torch.save({'losses': losses, 'params': params, 'shadow': shadow, 'buckets': n_overlapped},
             os.path.join(outdir, f'rank{rank}.pt'))
  dist.barrier()
  dist.destroy_process_group()


def gpu_worker(rank, world, port, outdir, family='wide', steps=4, global_batch=8):
  """One rank of a world_size-2 job whose ranks SHARE cuda:0 (the test boxes have one GPU): gloo process group (RCCL
  cannot put two ranks on one device), the product's HIP engine, gradient buckets all-reduced from inside a backward that
  is replayed as one hipGraph per bucket segment.  The exchange-after-backward run of the same steps must agree bit for
  bit; the parent test compares with one process on the whole batch."""
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  os.environ['STK_DDP_BUCKET_MB'] = '6'                 # several buckets for the few-M-parameter model
  dist.init_process_group('gloo', rank=rank, world_size=world)
  import soft_truncation_amd as st
  lib = st.engine.lib.load()                             # libstk.so; raises if it is missing
  assert lib.is_device
  calls = []
  real = dist.all_reduce
  dist.all_reduce = lambda *a, **k: (calls.append(a[0].numel()), real(*a, **k))[1]
  assert st.losses.OVERLAP_EXCHANGE
  # (a) default engine: forward hipGraph, backward launched eagerly segment by segment with the weight gradients on the
  #     side stream (engine/executor.py)
  losses, params, shadow = run_steps(st, lib, rank, world, steps, global_batch, family=family, device='cuda:0')
  n_overlapped, replays = len(calls), run_steps.graph_replays
  # (b) the same with the backward replayed as one hipGraph per bucket segment
  os.environ['STK_BWD_GRAPH'] = '1'
  del calls[:]
  losses_g, params_g, shadow_g = run_steps(st, lib, rank, world, steps, global_batch, family=family, device='cuda:0')
  n_graph, replays_g = len(calls), run_steps.graph_replays
  del os.environ['STK_BWD_GRAPH']
  # (c) exchange after the backward
  st.losses.OVERLAP_EXCHANGE = False
  del calls[:]
  losses_b, params_b, shadow_b = run_steps(st, lib, rank, world, steps, global_batch, family=family, device='cuda:0')
  dist.all_reduce = real
  assert n_overlapped >= steps * 3 and n_graph >= steps * 3, (n_overlapped, n_graph)   # >= 3 buckets per step left from inside the backward
  assert replays >= steps - 2, replays                  # the forward graph
  assert replays_g >= (steps - 2) * 4, replays_g        # forward + >= 3 backward segments, captured at step 2, replayed after
  assert torch.equal(params, params_b) and torch.equal(shadow, shadow_b) and torch.equal(losses, losses_b)
  assert torch.equal(params, params_g) and torch.equal(shadow, shadow_g) and torch.equal(losses, losses_g)
  torch.save({'losses': losses, 'params': params, 'shadow': shadow, 'buckets': n_overlapped, 'replays': replays},
             os.path.join(outdir, f'rank{rank}.pt'))
  dist.barrier()
  dist.destroy_process_group()
