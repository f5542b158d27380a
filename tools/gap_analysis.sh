#!/bin/bash
# Runs on the GPU box: kernel trace of a training-only bench run; reports busy time vs wall time and the idle gaps
# between consecutive kernels (launch bubbles inside the hipGraph replay).  usage: tools/gap_analysis.sh [bench args]
set -u
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gap
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --sampler-steps 0 --prof-steps 0 "$@" > /tmp/gap.log 2> /tmp/gap.err
F=$(find /tmp/gap -name "*kernel_trace.csv" | head -1)
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out"
OUT="$GRAFT_REPO_ROOT/gpurun_out/gaps.txt"
grep -h "^{" /tmp/gap.log | cut -c1-200 > "$OUT"
[ -n "$F" ] || { echo "no trace" >> "$OUT"; tail -5 /tmp/gap.err >> "$OUT"; cat "$OUT"; exit 1; }
python - "$F" >> "$OUT" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:70]) for r in rows))
# the timed region = the last 40% of the trace (steady-state graph replays)
t0 = ev[int(len(ev) * 0.6)][0]
ev = [e for e in ev if e[0] >= t0]
wall = ev[-1][1] - ev[0][0]
busy = sum(e[1] - e[0] for e in ev)
gaps = [(ev[i + 1][0] - ev[i][1], ev[i][2], ev[i + 1][2]) for i in range(len(ev) - 1)]
pos = [g for g in gaps if g[0] > 0]
print(f'kernels {len(ev)}  wall {wall/1e6:.2f} ms  sum of kernel durations {busy/1e6:.2f} ms  idle gaps {sum(g[0] for g in pos)/1e6:.2f} ms  overlapped {sum(-g[0] for g in gaps if g[0] < 0)/1e6:.2f} ms')
hist = collections.Counter()
for g in pos:
  b = 1
  while b < g[0] / 1000: b *= 2
  hist[b] += 1
print('gap histogram (us bucket upper bound: count):', sorted(hist.items()))
by = collections.defaultdict(lambda: [0, 0])
for g in pos:
  by[g[2]][0] += g[0]; by[g[2]][1] += 1
print('idle time in front of kernel (top 25):')
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
  print(f'  {k:<72} {n:6d} gaps  {t/1e6:8.3f} ms  avg {t/n/1e3:6.2f} us')
PY
cat "$OUT"
