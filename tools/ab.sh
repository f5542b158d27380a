#!/bin/bash
# A/B on ONE box (boxes differ by +-3 %): the bench step time under the engine's ablation switches.
# usage: tools/ab.sh "<ENV=..>" "<ENV=..>" ...   ("-" = defaults)
cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  E=""; [ "$V" != "-" ] && E="$V"
  for rep in 1 2; do
    R=$(env $E python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-probe --no-kernel-timer --prof-steps 0 --sampler-steps 0 --no-exchange-proxy 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))")
    echo "$V : $R"
  done
done
