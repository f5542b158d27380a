#!/bin/bash
# Runs on the GPU box: rocprofv3 per-kernel stats of one bench.py command line, top rows to gpurun_out/<name>.txt
# usage: tools/kernel_stats.sh <name> <bench args...>
set -u
NAME=$1; shift
TAG=$(echo "$NAME" | tr "/" "_")      # NAME may contain a directory (gpurun_out/<dir>/<name>.txt)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -o t -- python "$GRAFT_REPO_ROOT/bench.py" "$@" > /tmp/ks_$TAG.log 2> /tmp/ks_$TAG.err
F=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
mkdir -p "$(dirname "$GRAFT_REPO_ROOT/gpurun_out/$NAME")"
OUT="$GRAFT_REPO_ROOT/gpurun_out/$NAME.txt"
grep -h "^{" /tmp/ks_$TAG.log | cut -c1-260 > "$OUT"
if [ -n "$F" ]; then
  python - "$F" >> "$OUT" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
  name = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
  name = re.sub(r'\(.*$', '', name)[:100]
  print(f"{name:<100} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}%")
PY
else
  echo "no kernel_stats.csv" >> "$OUT"; tail -5 /tmp/ks_$TAG.err >> "$OUT"
fi
cat "$OUT"
