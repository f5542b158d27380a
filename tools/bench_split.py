#!/usr/bin/env python
"""Feasibility probe: do two half-batch evaluations (forward + backward, hipGraphs) on two HIP streams finish sooner than one
full-batch evaluation?  Two separate model replicas stand in for the two halves (no shared buffers).  Development tool.
usage: STK_WGRAD_STREAM=0 python tools/bench_split.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import soft_truncation_amd as st

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
cfg = st.configs.get_config('cifar10_ddpmpp_nll_st')
cfg.device = dev
sde = st.sde_lib.get_sde(cfg, None)


def make():
  m = st.models.utils.create_model(cfg, sde)
  m.module.engine().ensure_flat()
  m.train()
  return m


def run(models, batches, streams):
  outs = []
  for m, (x, t, go), s in zip(models, batches, streams):
    with torch.cuda.stream(s):
      outs.append((m(x, t) * go).sum())
  for o, s in zip(outs, streams):
    with torch.cuda.stream(s):
      o.backward()


def bench(B, n):
  models = [make() for _ in range(n)]
  g = torch.Generator().manual_seed(0)
  batches = [(torch.randn(B, 3, 32, 32, generator=g).to(dev), (torch.rand(B, generator=g) * 999).to(dev),
              torch.randn(B, 3, 32, 32, generator=g).to(dev)) for _ in range(n)]
  streams = [torch.cuda.Stream() for _ in range(n)]
  for s in streams:
    s.wait_stream(torch.cuda.current_stream())
  for _ in range(4):
    run(models, batches, streams)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    run(models, batches, streams)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  print(f'{n} x batch {B}: {dt * 1e3:.2f} ms per forward + backward of {n * B} images', flush=True)
  del models
  torch.cuda.empty_cache()


bench(128, 1)
bench(64, 2)
bench(64, 1)
bench(32, 4)
