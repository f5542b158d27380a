#!/usr/bin/env python
"""Effective shader clock per kernel from ONE rocprofv3 run that collects GRBM_GUI_ACTIVE with the kernel trace
(MI355X_MICROARCH.md "DVFS give-back": effective clock = GRBM_GUI_ACTIVE / kernel wall time).

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d DIR -o r -- <cmd>
    python tools/kernel_clock.py DIR

GRBM_GUI_ACTIVE is reported summed over the XCDs that were active; the script prints the raw ratio and the ratio / 8."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
cc = glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)
kt = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
if not cc:
  sys.exit('no counter_collection.csv under ' + root)
dur = {}
if kt:
  for r in csv.DictReader(open(kt[0])):
    dur[r.get('Dispatch_Id') or r.get('Correlation_Id')] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc[0])):
  name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])
  name = re.sub(r'\(.*$', '', name)[:80]
  agg[name][r['Counter_Name']] += float(r['Counter_Value'])
  did = r.get('Dispatch_Id') or r.get('Correlation_Id')
  if did not in seen[name]:
    seen[name].add(did)
    if 'Start_Timestamp' in r and r['Start_Timestamp']:
      agg[name]['_ns'] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    elif did in dur:
      agg[name]['_ns'] += dur[did]
rows = sorted(agg.items(), key=lambda kv: -kv[1].get('_ns', 0))
print(f"{'kernel':<80} {'calls':>6} {'avg_us':>8} {'GUI_ACTIVE/ns':>14} {'/8 (GHz)':>9} {'mfma busy of 4':>15}")
for name, c in rows[:40]:
  n = len(seen[name])
  ns = c.get('_ns', 0.0)
  if ns <= 0:
    continue
  ratio = c.get('GRBM_GUI_ACTIVE', 0.0) / ns
  busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CU_CYCLES'] if c.get('SQ_BUSY_CU_CYCLES') else float('nan')
  print(f'{name:<80} {n:6d} {ns / n / 1e3:8.1f} {ratio:14.3f} {ratio / 8:9.3f} {busy:15.3f}')
