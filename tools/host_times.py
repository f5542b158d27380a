#!/usr/bin/env python
"""Host-side time of the phases of a training step (runs on the GPU box): how long the host spends INSIDE the hipGraph
replays of the forward / backward (a large graph launch blocks the calling thread while its nodes are queued) and in the
rest of step_fn.  usage: python tools/host_times.py [steps] [workload]
("stream already idle when the loss-copy wait returned" N of N times = the HOST is the bottleneck of the step: the batch-4 regime)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import soft_truncation_amd as st
from importlib import import_module

ex_mod = import_module('soft-truncation_amd.engine.executor')
acc = {'fwd': 0.0, 'bwd': 0.0, 'n': 0}
orig = ex_mod.Executor._replay


def timed(self, c, direction, *a, **k):
  t0 = time.perf_counter()
  r = orig(self, c, direction, *a, **k)
  acc[direction] += time.perf_counter() - t0
  return r


ex_mod.Executor._replay = timed
acc['evsync'] = 0.0
_orig_sync = torch.cuda.Event.synchronize


def _timed_sync(self):
  t0 = time.perf_counter()
  r = _orig_sync(self)
  acc['evsync'] += time.perf_counter() - t0
  acc['drained'] = acc.get('drained', 0) + int(torch.cuda.current_stream().query())      # is the whole queue done already?
  acc['syncs'] = acc.get('syncs', 0) + 1
  return r


torch.cuda.Event.synchronize = _timed_sync
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
device = torch.device('cuda', 0)
torch.cuda.set_device(0)
cfg_name, B, desc = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else 'cifar10']
cfg = st.configs.get_config(cfg_name)
cfg.device = device
st.engine.ddp.seed_everything(cfg.seed)
sde = st.sde_lib.get_sde(cfg, None)
model = st.models.utils.create_model(cfg, sde)
opt = st.losses.get_optimizer(cfg, model.parameters())
ema = st.models.ema.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
state = dict(optimizer=opt, model=model, ema=ema, step=0)
step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
batch = st.datasets.synthetic_batch(cfg, B).to(device)
for _ in range(5):
  step_fn(state, batch)
torch.cuda.synchronize()
acc.update(fwd=0.0, bwd=0.0, evsync=0.0, drained=0, syncs=0)
host = 0.0
t_all = time.perf_counter()
for _ in range(steps):
  t0 = time.perf_counter()
  step_fn(state, batch)
  host += time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
print(f'steps {steps}: wall {wall / steps * 1e3:.2f} ms/step; host inside step_fn {host / steps * 1e3:.2f} ms/step, of which inside the '
      f'forward replay {acc["fwd"] / steps * 1e3:.2f}, the backward replay {acc["bwd"] / steps * 1e3:.2f}, waiting for the loss copy {acc["evsync"] / steps * 1e3:.2f}; '
      f'stream already idle when that wait returned: {acc["drained"]} of {acc["syncs"]} times')
