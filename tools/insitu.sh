#!/bin/bash
# In-situ A/B of engine switches: the training step of one bench.py workload under each environment setting, one line per
# setting (ms per step).  Kernel choices are made on in-step numbers only (DESIGN.md "Lessons": micro-benchmarks flatter every
# kernel that is not bound by the matrix pipe).  Replaces the ten single-purpose tools/_probe/insitu_sweep*.sh of round 5
# (their settings and results are recorded in profiles/r05_insitu_sweeps.txt).
#
#   tools/insitu.sh WORKLOAD "--steps 30 --warmup 5" "A=1" "STK_X2W_GROUPS=1" "STK_X2W_WGS=320 STK_KSPLIT_WGS=384" "STK_LIBSTK=/path/to/other/libstk.so" ...
#
# Run on the GPU box (gpurun); first and last setting should be the default ("A=1") so that drift of the box shows.
cd "$(dirname "$0")/.."
W="$1"; shift
ARGS="$1"; shift
for cfg in "$@"; do
  ms=$(timeout 300 env $cfg python bench.py --workload $W $ARGS --no-cpu-baseline --sampler-steps 0 --no-exchange-proxy --no-parity-probe \
        --no-kernel-timer --prof-steps 0 --no-extra-workloads --detail /tmp/insitu_detail.json 2>/tmp/insitu.err \
       | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)
  if [ -z "$ms" ]; then ms="FAILED: $(tail -c 300 /tmp/insitu.err | tr '\n' ' ')"; fi
  printf '%-12s %-60s %s\n' "$W" "[$cfg]" "$ms"
done
