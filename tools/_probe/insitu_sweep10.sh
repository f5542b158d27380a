#!/bin/bash
# the 256x256 net at batch 4 (BASELINE configs[4]): ~1800 small launches per backward -- eager two-stream backward against hipGraph replay
cd "$(dirname "$0")/../.."
one() {
  local W="$1"; shift; local A="$1"; shift
  timeout 250 env "$@" python bench.py --workload $W $A --no-cpu-baseline --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
}
for W in celebahq256 celeba64; do
  for cfg in "A=1" "STK_BWD_GRAPH=1" "STK_WGRAD_STREAM=0" "STK_WGRAD_STREAM=0 STK_X2W_WGS_ALONE=512" "STK_X2W_WGS=320" "STK_X2W_WGS=192" "A=1"; do
    echo "$W [$cfg]  $(one $W '--steps 12 --warmup 4' $cfg)"
  done
done
