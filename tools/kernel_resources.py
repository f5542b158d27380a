#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py soft-truncation_amd/csrc/conv.hip [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
csrc = os.path.join(ROOT, 'soft-truncation_amd', 'csrc')
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950',
                      '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
  m = re.search(r'Function Name: (\S+)', line)
  if m:
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    cur = {'name': re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0][:110]}
    rows.append(cur)
    continue
  m = re.search(r'remark:\s+([A-Za-z][^:]*): (\d+)', line)
  if m and cur is not None:
    cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
  if flt in r['name']:
    print(f"{r['name']:<110} V{r.get('VGPRs', 0):>4} A{r.get('AGPRs', 0):>4} S{r.get('TotalSGPRs', 0):>4} "
          f"spillV{r.get('VGPRs Spill', 0):>3} spillS{r.get('SGPRs Spill', 0):>3} occ{r.get('Occupancy [waves/SIMD]', 0):>2} "
          f"LDS{r.get('LDS Size [bytes/block]', 0):>7}")
