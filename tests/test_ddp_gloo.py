"""Multi-GPU path on CPU: world_size 2 over gloo.  One process per rank, replicas with a bucketed gradient
all-reduce (engine/ddp.py), a shared numpy stream for t_min, per-rank noise.  The check: two ranks training on
the two halves of a global batch reach the same parameters (to round-off) as one process on the whole
batch -- the property the 1/2/4/8-GPU scaling runs rely on."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import _ddp_worker as W


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def test_bucket_ranges(st):
  r = st.engine.ddp.bucket_ranges(10, 4)
  assert r == [(0, 4), (4, 8), (8, 10)]
  assert st.engine.ddp.bucket_ranges(0, 4) == []


@pytest.mark.timeout(600)
def test_two_ranks_match_single_process(st, ref_lib, tmp_path):
  world = 2
  port = _free_port()
  ctx = mp.get_context('spawn')
  procs = [ctx.Process(target=W.worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(500)
    assert p.exitcode == 0, f'rank exited with {p.exitcode}'
  got = [torch.load(os.path.join(str(tmp_path), f'rank{r}.pt')) for r in range(world)]
  # replicas stay identical
  assert torch.equal(got[0]['params'], got[1]['params'])
  assert torch.equal(got[0]['shadow'], got[1]['shadow'])
  # single process, whole batch, same noise
  losses, params, shadow = W.run_steps(st, ref_lib, 0, 1, steps=2, global_batch=4)
  # per-sample losses: step i of the single run = [rank0 half, rank1 half]
  for i in range(2):
    both = torch.cat([got[0]['losses'][2 * i:2 * i + 2], got[1]['losses'][2 * i:2 * i + 2]])
    assert torch.allclose(both, losses[4 * i:4 * i + 4], rtol=1e-4, atol=0)
  lr = 2e-4
  assert (got[0]['params'] - params).abs().max().item() <= 0.05 * lr * 2
  assert (got[0]['shadow'] - shadow).abs().max().item() <= 0.05 * lr * 2


def test_backward_segments_cover_the_gradient_buffer(st, ref_lib):
  """Program.backward_segments: the buckets partition [0, n_train), come top-down, and a bucket is only released after
  the last op that writes into it."""
  import torch
  from _model_util import build_pair, tiny_config
  cfg, cfg_cpu, sde, model, ref = build_pair(st, tiny_config(st, 'vp'), ref_lib)
  x = torch.randn(2, 3, 16, 16)
  model(x, torch.rand(2) * 999).sum().backward()
  ex = model.module.engine()
  prog = next(iter(ex.programs.values()))
  n = ex.flat.n_train
  segs = prog.backward_segments(n, 20000)
  ranges = [r for _, rs in segs for r in rs]
  assert len(ranges) >= 3
  assert ranges[0][1] == n and ranges[-1][0] == 0
  assert all(a[0] == b[1] for a, b in zip(ranges, ranges[1:]))          # contiguous, descending
  ends = [e for e, _ in segs]
  assert ends == sorted(ends) and ends[-1] == len(prog.graph.ops)
  # no op after a bucket's release point touches a parameter inside it
  from importlib import import_module
  Tensor = import_module('soft-truncation_amd.engine.graph').Tensor
  order = list(reversed(prog.graph.ops))
  for end, rs in segs:
    for op in order[end:]:
      for v in vars(op).values():
        if isinstance(v, Tensor) and v.space == 'param' and v.goff is not None:
          assert not any(lo <= v.goff < hi for lo, hi in rs), (type(op).__name__, v.goff, rs)
