#!/usr/bin/env python
"""Idle time inside steady-state training steps from a rocprofv3 kernel trace (runs on the GPU box after
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 8 --warmup 2 ...`): a step = the window
between two consecutive adam_kernel launches; prints busy / idle per step and the largest gaps with their neighbours.
usage: python tools/step_gaps.py DIR"""
import csv
import glob
import os
import re
import sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']),
             re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0][:60]) for r in csv.DictReader(open(f)))
adam = [i for i, e in enumerate(ev) if e[2].startswith('adam_kernel')]
print('adam launches', len(adam))
for a, b in list(zip(adam, adam[1:]))[-5:]:
  w = ev[a:b + 1]
  wall = w[-1][0] - w[0][0]
  busy = sum(e[1] - e[0] for e in w[:-1])
  gaps = sorted(((w[i + 1][0] - w[i][1], w[i][2], w[i + 1][2]) for i in range(len(w) - 1)), reverse=True)
  print(f'step: wall {wall / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(wall - busy) / 1e6:.3f} ms  kernels {len(w) - 1}')
  for g in gaps[:8]:
    print(f'    {g[0] / 1e3:8.1f} us between {g[1]}  ->  {g[2]}')

# context of the largest gap of the last step: the kernels around it with their start offsets
a, b = adam[-2], adam[-1]
w = ev[a:b + 1]
gi = max(range(len(w) - 1), key=lambda i: w[i + 1][0] - w[i][1])
print('around the largest gap of the last step (start offset us, duration us, kernel):')
for e in w[max(0, gi - 8):gi + 6]:
  print(f'    {(e[0] - w[gi][1]) / 1e3:10.1f} {(e[1] - e[0]) / 1e3:8.1f}  {e[2]}')
