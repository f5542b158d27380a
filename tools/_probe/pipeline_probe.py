#!/usr/bin/env python
"""Probe (timing only): would pipelining TWO half batches through the training step pay -- the forward of half B on a third stream
beside the backward of half A -- so that the forward's HBM-bound GroupNorm passes and the backward's sit beside the other half's GEMMs?

  variant full      one step_fn call on the whole batch (the benched step)
  variant serial    the two halves one after the other (what the kernels lose at half the batch)
  variant pipelined forward(A) | forward(B) on stream S2 beside backward(A) | backward(B)

The pipelined variant shares the program's K-split workspace between forward(B) and backward(A), so its NUMBERS ARE NOT VALID
RESULTS -- only its time is of interest.  usage: python tools/_probe/pipeline_probe.py [workload] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
import soft_truncation_amd as st
from importlib import import_module

ex_mod = import_module('soft-truncation_amd.engine.executor')
workload = sys.argv[1] if len(sys.argv) > 1 else 'cifar10'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
device = torch.device('cuda', 0)
torch.cuda.set_device(0)
cfg_name, B, desc = bench.WORKLOADS[workload]
cfg = st.configs.get_config(cfg_name)
cfg.device = device
st.engine.ddp.seed_everything(cfg.seed)
sde = st.sde_lib.get_sde(cfg, None)
state, step_fn = bench.build_training(st, cfg, sde)
model, optimizer = state['model'], state['optimizer']
optimize_fn = st.losses.optimization_manager(cfg)
loss_fn = st.losses._pick_loss_fn(cfg, sde, True)
batch = st.datasets.synthetic_batch(cfg, B, device=device, generator=torch.Generator().manual_seed(1))
main = torch.cuda.current_stream(device)
side = ex_mod.checked_side_stream(device)
S2 = None
for _ in range(40):
  cand = torch.cuda.Stream(device)
  if cand.cuda_stream in (main.cuda_stream, side.cuda_stream):
    continue
  if ex_mod._overlap_ratio(main, cand) < 1.5 and ex_mod._overlap_ratio(side, cand) < 1.5:
    S2 = cand
    break
print('third stream found:', S2 is not None)
half = B // 2


def split_step(pipelined):
  optimizer.zero_grad()
  t_min = sde.get_t_min(cfg)
  A, Bh = batch[:half], batch[half:]
  kw = dict(importance_sampling=cfg.training.importance_sampling, t_min=t_min)
  lA = loss_fn(model, A, **kw)
  if pipelined:
    eA = torch.cuda.Event()
    eA.record(main)
    S2.wait_event(eA)
    with torch.cuda.stream(S2):
      lB = loss_fn(model, Bh, **kw)
      eB = torch.cuda.Event()
      eB.record(S2)
    (0.5 * torch.mean(lA)).backward()
    main.wait_event(eB)
    (0.5 * torch.mean(lB)).backward()
  else:
    (0.5 * torch.mean(lA)).backward()
    lB = loss_fn(model, Bh, **kw)
    (0.5 * torch.mean(lB)).backward()
  optimize_fn(optimizer, model.parameters(), step=state['step'])
  state['step'] += 1
  state['ema'].update(model.parameters())


def timeit(fn, n):
  for _ in range(6):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  torch.cuda.synchronize()
  return 1e3 * (time.perf_counter() - t0) / n


print(f'{workload}: full       {timeit(lambda: step_fn(state, batch), steps):8.3f} ms per step')
print(f'{workload}: serial     {timeit(lambda: split_step(False), steps):8.3f} ms per step')
if S2 is not None:
  print(f'{workload}: pipelined  {timeit(lambda: split_step(True), steps):8.3f} ms per step   (timing only)')
print(f'{workload}: full       {timeit(lambda: step_fn(state, batch), steps):8.3f} ms per step')
