#!/usr/bin/env python
"""Per-op GPU time of one training step's network evaluation (runs on the GPU box): every op of the planned graph is
bracketed by HIP events in its forward and in its backward (eager launches, STK_GRAPHS=0), grouped by op class and
shape.  An op = all launches of one graph node (e.g. a convolution's backward = bias gradient + planes of dy + data
gradient + weight gradient + their reduces).  usage: python tools/op_times.py [--workload cifar10] [--batch N] [--reps R]"""
import argparse
import collections
import json
import os
import sys

os.environ['STK_GRAPHS'] = '0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from importlib import import_module

import bench
import soft_truncation_amd as st

graph_mod = import_module('soft-truncation_amd.engine.graph')


def label(op):
  n = type(op).__name__
  if isinstance(op, graph_mod.Conv):
    return (f'Conv{op.KH}x{op.KW} {op.C1}+{op.C2}->{op.Cout} @{op.H}x{op.W}' + (' s2' if op.stride != 1 else '') +
            (' pl' if op.pl_fwd else '') + (' +res' if op.res is not None else '') + (' +temb' if op.temb is not None else ''))
  if isinstance(op, graph_mod.GroupNormAct):
    return f'GroupNorm {op.C1}+{op.C2} @{op.HW} act={op.act}' + (' planes' if op.y.pl_maker is op else '') + \
        ('' if op.y.f32_fwd or op.y.f32_bwd else ' nof32')
  y = getattr(op, 'y', None)
  return n + (' ' + 'x'.join(str(s) for s in y.shape) if y is not None else '')


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--workload', default='cifar10')
  ap.add_argument('--batch', type=int, default=0)
  ap.add_argument('--reps', type=int, default=3)
  ap.add_argument('--out', default='')
  args = ap.parse_args()
  device = torch.device('cuda', 0)
  torch.cuda.set_device(0)
  cfg_name, B, desc = bench.WORKLOADS[args.workload]
  B = args.batch or B
  cfg = st.configs.get_config(cfg_name)
  cfg.device = device
  st.engine.ddp.seed_everything(cfg.seed)
  sde = st.sde_lib.get_sde(cfg, None)
  model = st.models.utils.create_model(cfg, sde)
  model.module.engine().ensure_flat()
  optimizer = st.losses.get_optimizer(cfg, model.parameters())
  ema = st.models.ema.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
  state = dict(optimizer=optimizer, model=model, ema=ema, step=0)
  step_fn = st.losses.get_step_fn(cfg, sde, train=True, optimize_fn=st.losses.optimization_manager(cfg))
  batch = st.datasets.synthetic_batch(cfg, B, device=device, generator=torch.Generator().manual_seed(1234))
  for _ in range(2):
    step_fn(state, batch)
  torch.cuda.synchronize()
  ex = model.module.engine()
  records = []
  for prog in ex.programs.values():
    for op in prog.graph.ops:
      for direction in ('forward', 'backward'):
        def wrap(fn, op=op, direction=direction):
          def timed(rt):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); fn(rt); e.record()
            records.append((label(op), direction, s, e, getattr(op, 'flops', 0.0)))
          return timed
        setattr(op, direction, wrap(getattr(op, direction)))
  t_all = []
  for _ in range(args.reps):
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); step_fn(state, batch); e.record()
    torch.cuda.synchronize()
    t_all.append(s.elapsed_time(e))
  agg = collections.OrderedDict()
  for lab, d, s, e, fl in records:
    a = agg.setdefault((lab, d), [0, 0.0, 0.0])
    a[0] += 1; a[1] += s.elapsed_time(e); a[2] += fl
  rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
  tot = sum(v[1] for v in agg.values()) / args.reps
  lines = [f'# {desc}, batch {B}: per-op GPU time of the network evaluation inside one training step (eager launches, '
           f'HIP events per op, mean of {args.reps} steps); eager step {sum(t_all) / len(t_all):.2f} ms, ops total {tot:.2f} ms',
           f'{"op":<58} {"dir":<9} {"n/step":>6} {"avg_us":>9} {"ms/step":>8} {"TF/s":>7}']
  for (lab, d), (n, ms, fl) in rows:
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 and fl > 0 else 0.0
    lines.append(f'{lab:<58} {d:<9} {n / args.reps:6.0f} {1e3 * ms / n:9.1f} {ms / args.reps:8.3f} {tf:7.1f}')
  by_class = collections.OrderedDict()
  for (lab, d), (n, ms, fl) in rows:
    k = (lab.split(' ')[0], d)
    by_class[k] = by_class.get(k, 0.0) + ms / args.reps
  lines.append('')
  for k, v in sorted(by_class.items(), key=lambda kv: -kv[1]):
    lines.append(f'{k[0]:<20} {k[1]:<9} {v:8.3f} ms/step')
  text = '\n'.join(lines)
  print(text)
  if args.out:
    open(args.out, 'w').write(text + '\n')


if __name__ == '__main__':
  main()
