#!/bin/bash
cd "$(dirname "$0")/../.."
one() {  # workload, steps args, env...
  local W="$1"; shift; local A="$1"; shift
  env "$@" python bench.py --workload $W $A --no-cpu-baseline --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
}
CANDS=("A=1" "STK_SIDE_SHORTCUT=0 STK_X2W_WGS=256" "STK_SIDE_SHORTCUT=0 STK_X2W_WGS=256 STK_W1_WGS=128" "STK_SIDE_SHORTCUT=0 STK_X2W_WGS=320" "STK_SIDE_SHORTCUT=0 STK_X2W_WGS=320 STK_W1_WGS=128" "STK_SIDE_SHORTCUT=0 STK_X2W_GROUPS=1 STK_X2W_WGS=192" "STK_SIDE_SHORTCUT=0 STK_X2W_GROUPS=3 STK_X2W_WGS=320 STK_W1_WGS=128")
for rep in 1 2 3; do
  for c in "${CANDS[@]}"; do echo "cifar10 r$rep [$c]  $(one cifar10 "" $c)"; done
done
for c in "${CANDS[@]}"; do echo "celeba64 [$c]  $(one celeba64 "--steps 12 --warmup 4" $c)"; done
for c in "${CANDS[@]}"; do echo "celebahq256 [$c]  $(one celebahq256 "--steps 12 --warmup 4" $c)"; done
