#!/usr/bin/env python
"""Two-stream timeline of steady-state training steps from a rocprofv3 kernel trace (round 5).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline ...
    python tools/stream_timeline.py DIR

A step = the window between two consecutive adam_kernel launches.  Per step and hardware queue: busy time; for the whole step:
time with no kernel running, with exactly one, with two or more (the weight-gradient stream really beside the main chain), and how
long the side queue still works after the main queue's last backward kernel (the join at the end of the backward)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r[qkey],
             re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0][:60]) for r in rows)
adam = [i for i, e in enumerate(ev) if e[3].startswith('adam_kernel')]
print('queue key', qkey, ' adam launches', len(adam))
for a, b in list(zip(adam, adam[1:]))[-4:]:
  w = ev[a + 1:b + 1]
  t0, t1 = ev[a][1], ev[b][0]
  busy = defaultdict(int)
  for s, e, q, n in w:
    busy[q] += e - s
  # sweep: number of kernels running
  pts = sorted([(s, 1) for s, e, q, n in w] + [(e, -1) for s, e, q, n in w])
  level, last, hist = 0, t0, defaultdict(int)
  for t, d in pts:
    hist[min(level, 2)] += max(t - last, 0)
    last = max(t, last)
    level += d
  main = max(busy, key=busy.get)
  side_end = max((e for s, e, q, n in w if q != main), default=t0)
  main_bwd_end = max(e for s, e, q, n in w if q == main and not n.startswith(('adam', 'sumsq', 'ema')))
  first_side = min((s for s, e, q, n in w if q != main and n.startswith(('x2w', 'x2::wg', 'splitk'))), default=t0)
  print(f'step {(t1 - t0) / 1e6:7.3f} ms | idle {hist[0] / 1e6:6.3f}  one kernel {hist[1] / 1e6:6.3f}  two or more {hist[2] / 1e6:6.3f} | '
        + '  '.join(f'queue {q}: {v / 1e6:6.3f} ms' for q, v in sorted(busy.items(), key=lambda kv: -kv[1]))
        + f' | forward+loss until first side launch {(first_side - t0) / 1e6:6.3f} | side queue ends {(side_end - main_bwd_end) / 1e6:+.3f} ms after the main chain')
# what runs on the side queue while the main queue is idle-waiting at the join (last step)
a, b = adam[-2], adam[-1]
w = ev[a + 1:b + 1]
busy = defaultdict(int)
for s, e, q, n in w:
  busy[q] += e - s
main = max(busy, key=busy.get)
main_bwd_end = max(e for s, e, q, n in w if q == main and not n.startswith(('adam', 'sumsq', 'ema')))
print('side-queue kernels ending after the main chain (last step):')
for s, e, q, n in w:
  if q != main and e > main_bwd_end:
    print(f'   +{(s - main_bwd_end) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  {n}')
# where the chip is idle (no kernel on any queue), last step: by the kernel that ends the gap
pts = sorted([(s, 1, q, n) for s, e, q, n in w] + [(e, -1, q, n) for s, e, q, n in w])
level, gap_start, prev = 0, None, ''
agg = defaultdict(lambda: [0, 0])
big = []
for t, d, q, n in pts:
  if d == 1:
    if level == 0 and gap_start is not None and t > gap_start:
      key = ('main' if q == main else 'side', n)
      agg[key][0] += t - gap_start; agg[key][1] += 1
      big.append((t - gap_start, prev, ('main ' if q == main else 'side ') + n, (gap_start - w[0][0]) / 1e6))
    level += 1
  else:
    level -= 1
    if level == 0:
      gap_start, prev = t, n
print('idle time in front of (queue, kernel), last step:')
for (qq, n), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
  print(f'   {qq} {n:<62} {c:5d} gaps {t / 1e3:9.1f} us  avg {t / c / 1e3:6.2f}')
print('largest single gaps (us, after kernel -> before kernel, at ms into the step):')
for g in sorted(big, reverse=True)[:12]:
  print(f'   {g[0] / 1e3:8.1f}  {g[1]:<50} -> {g[2]:<56} @ {g[3]:.2f}')
