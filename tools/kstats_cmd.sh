#!/bin/bash
# Runs on the GPU box: rocprofv3 per-kernel stats of an arbitrary command, top rows to stdout.
# usage: tools/kstats_cmd.sh [-n rows] <command...>
ROWS=25
if [ "$1" = "-n" ]; then ROWS=$2; shift 2; fi
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=/tmp/kstats_$$
cd /tmp && export TMPDIR=/tmp
( cd "$R" && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- "$@" > $D.log 2> $D.err )
F=$(find $D -name "*kernel_stats.csv" | head -1)
if [ -z "$F" ]; then echo "no kernel_stats.csv"; tail -5 $D.err; exit 1; fi
python - "$F" $ROWS <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
  name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Name'])
  name = re.sub(r'\(.*$', '', name)[:90]
  print(f"{name:<90} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}%")
PY
