#!/usr/bin/env python
"""One sampler network evaluation of a bench.py workload, repeated: ms per evaluation (and, under tools/kstats_cmd.sh, where the
time goes).  The PC sampler of BASELINE configs[4] (NCSN++ 256x256, sampling batch 16) evaluates the score network 2 N + 1 times
on fixed weights; this is that evaluation alone -- forward only, eval mode, weights prepared once (Executor.frozen_weights), the
inference program replayed as a hipGraph -- without the predictor / corrector arithmetic around it.

    python tools/sampler_eval.py --workload celebahq256 --batch 16 --evals 20
    tools/kstats_cmd.sh -n 40 python tools/sampler_eval.py --workload celebahq256 --batch 16 --evals 20
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import soft_truncation_amd as st

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='celebahq256', choices=sorted(bench.WORKLOADS))
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--evals', type=int, default=20)
args = ap.parse_args()
cfg_name, _, desc = bench.WORKLOADS[args.workload]
cfg = st.configs.get_config(cfg_name)
device = torch.device('cuda', 0)
cfg.device = device
sde = st.sde_lib.get_sde(cfg, None)
torch.manual_seed(0)
model = st.models.utils.create_model(cfg, sde)
model.eval()
score_fn = st.models.utils.get_score_fn(cfg, sde, model, train=False, continuous=cfg.training.continuous)
S = cfg.data.image_size
x = torch.randn(args.batch, cfg.data.num_channels, S, S, device=device)
t = torch.full((args.batch,), 0.5, device=device)
with torch.no_grad(), model.module.engine().frozen_weights():
  for _ in range(3):
    score_fn(x, t)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.evals):
    score_fn(x, t)
  torch.cuda.synchronize()
  ms = 1e3 * (time.perf_counter() - t0) / args.evals
flops = {'cifar10': 21.693e9, 'imagenet32': 21.693e9, 'celeba64': 84.104e9, 'celebahq256': 533.437e9}[args.workload]   # forward FLOPs / image, SURVEY 8(d)
tf = flops * args.batch / (ms * 1e-3) / 1e12
print(f'{desc}: batch {args.batch}, {args.evals} evaluations, {ms:.3f} ms per evaluation = {tf:.1f} TFLOP/s fp32-equivalent = '
      f'{tf / bench.PEAK_X2_TFLOPS:.3f} of {bench.PEAK_X2_TFLOPS:.0f}')
