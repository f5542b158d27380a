#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (sum over dispatches)."""
import collections
import csv
import re
import sys

path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(path)):
  name = re.sub(r'\(anonymous namespace\)::|igemm::|void ', '', r['Kernel_Name'])
  name = re.sub(r'Cfg<(\d+), (\d+), (\d+)>', r'C\1x\3', name)[:70]
  agg[name][r['Counter_Name']] += float(r['Counter_Value'])
  cnt[(name, r['Counter_Name'])] += 1
for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
  print(name)
  print('   ', {k: f'{v:.3g}' for k, v in c.items()})
  if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CU_CYCLES' in c:
    print('    mfma busy / cu busy = %.3f' % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(c['SQ_BUSY_CU_CYCLES'], 1)))
