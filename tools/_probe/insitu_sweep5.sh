#!/bin/bash
cd "$(dirname "$0")/../.."
run() {
  local label="$1"; shift
  local ms=$(env "$@" python bench.py --no-cpu-baseline --no-extra-workloads --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$label  $ms"
}
run "default                 " A=1
for G in 3 1 2; do for W in 128 192 256 320 384; do
run "SS=0 G=$G WGS=$W          " STK_SIDE_SHORTCUT=0 STK_X2W_GROUPS=$G STK_X2W_WGS=$W
done; done
run "default                 " A=1
run "SS=1 G=3 WGS=320        " STK_X2W_GROUPS=3 STK_X2W_WGS=320
run "SS=0 G=3 WGS=320 MINCH=6" STK_SIDE_SHORTCUT=0 STK_X2W_GROUPS=3 STK_X2W_WGS=320 STK_KSPLIT_MINCH=6
run "SS=0 G=3 WGS=320 W1=128 " STK_SIDE_SHORTCUT=0 STK_X2W_GROUPS=3 STK_X2W_WGS=320 STK_W1_WGS=128
for cfg in "A=1" "STK_SIDE_SHORTCUT=0" "STK_X2W_GROUPS=3" "STK_X2W_WGS=320" "STK_SIDE_SHORTCUT=0 STK_X2W_WGS=320" "STK_SIDE_SHORTCUT=0 STK_X2W_GROUPS=1 STK_X2W_WGS=256"; do
  ms=$(env $cfg python bench.py --workload celeba64 --steps 12 --warmup 4 --no-cpu-baseline --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "celeba64 [$cfg]  $ms"
done
