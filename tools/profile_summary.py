#!/usr/bin/env python
"""Summarise one tools/profile_round.sh run: per-kernel time (rocprofv3 --kernel-trace --stats), per-launch HBM
traffic of each kernel from the FETCH_SIZE / WRITE_SIZE counter passes (MI355X_MICROARCH.md: separate --pmc passes,
unit KB, FETCH_SIZE on gfx950 reports half of a wide coalesced read -> doubled here), MFMA-busy fraction."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
extra = sys.argv[2] if len(sys.argv) > 2 else ''      # e.g. '--workload celeba64' (tools/profile_workload.sh)


def short(n):
  n = re.sub(r'\(anonymous namespace\)::|igemm::|void |x3::', '', n)      # x2:: is kept: it tells the two gemm_kernel families apart
  n = re.sub(r'Cfg<(\d+), (\d+), (\d+)>', r'C\1', n)
  return n.split('(')[0][:90]


def first(pattern):
  g = glob.glob(os.path.join(root, pattern))
  return g[0] if g else None


def counters(sub):
  f = first(f'{sub}/*counter_collection.csv')
  agg = collections.defaultdict(lambda: collections.defaultdict(float))
  cnt = collections.Counter()
  if f:
    for r in csv.DictReader(open(f)):
      k = short(r['Kernel_Name'])
      agg[k][r['Counter_Name']] += float(r['Counter_Value'])
      cnt[(k, r['Counter_Name'])] += 1
  return agg, cnt


bench = json.loads(open(os.path.join(root, 'bench.json')).read().strip().splitlines()[-1])
print('bench:', json.dumps({k: bench[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'n_gpus', 'steps')}))
print('roofline:', json.dumps(bench.get('roofline')))

stats = first('stats/*kernel_stats.csv')
rows = list(csv.DictReader(open(stats))) if stats else []
steps = 13.0   # 10 timed + 3 warm-up steps in the profiled command (+ eager profiling steps, see the bench line)
print(f'\nper-kernel time, rocprofv3 --kernel-trace --stats of `STK_WGRAD_STREAM=0 python bench.py {extra} --steps 10 --warmup 3 --no-cpu-baseline --sampler-steps 0` (one stream, training only)')
print(f'{"kernel":92s} {"calls":>8s} {"avg_us":>9s} {"total_ms":>10s} {"%":>6s}')
for r in rows[:45]:
  print(f'{short(r["Name"]):92s} {int(r["Calls"]):8d} {float(r["AverageNs"]) / 1e3:9.1f} {float(r["TotalDurationNs"]) / 1e6:10.2f} {float(r["Percentage"]):6.2f}')

fetch, fc = counters('pmc_fetch')
write, wc = counters('pmc_write')
mfma, mc = counters('pmc_mfma')
print('\nper-launch HBM traffic (KB counters -> bytes; FETCH x2 per the gfx950 note; WRITE_SIZE uncalibrated) and MFMA busy')
print(f'{"kernel":92s} {"launches":>8s} {"fetch_MB":>9s} {"write_MB":>9s} {"mfma_busy/cu_busy(of 4)":>24s}')
keys = sorted(fetch, key=lambda k: -fetch[k].get('FETCH_SIZE', 0))
out = {}
for k in keys[:40]:
  n = fc[(k, 'FETCH_SIZE')]
  f_mb = fetch[k]['FETCH_SIZE'] * 1024 * 2 / max(n, 1) / 1e6
  w_mb = write[k].get('WRITE_SIZE', 0.0) * 1024 / max(wc[(k, 'WRITE_SIZE')], 1) / 1e6
  mb = mfma[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(mfma[k].get('SQ_BUSY_CU_CYCLES', 0.0), 1.0)
  out[k] = dict(launches=n, fetch_MB=f_mb, write_MB=w_mb, mfma_busy_per_cu=mb)
  print(f'{k:92s} {n:8d} {f_mb:9.2f} {w_mb:9.2f} {mb:24.3f}')
json.dump(out, open(os.path.join(root, 'traffic.json'), 'w'), indent=1)
