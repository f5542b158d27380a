#!/bin/bash
cd "$(dirname "$0")/../.."
run() {
  local label="$1"; shift
  local ms=$(env "$@" python bench.py --no-cpu-baseline --no-extra-workloads --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$label  $ms"
}
B="STK_SIDE_SHORTCUT=0 STK_X2W_GROUPS=3"
run "default                          " A=1
run "SS=0 G=3 WGS=320                 " $B STK_X2W_WGS=320
run "SS=0 G=3 WGS=384                 " $B STK_X2W_WGS=384
run "SS=0 G=3 WGS=448                 " $B STK_X2W_WGS=448
run "SS=0 G=3 WGS=512                 " $B STK_X2W_WGS=512
run "SS=0 G=3 WGS=640                 " $B STK_X2W_WGS=640
run "SS=0 G=3 WGS=768                 " $B STK_X2W_WGS=768
run "SS=0 G=2 WGS=384                 " STK_SIDE_SHORTCUT=0 STK_X2W_WGS=384
run "SS=0 G=3 WGS=384 SLAB=64         " $B STK_X2W_WGS=384 STK_WGRAD_SLAB_MB=64
run "SS=0 G=3 WGS=384 MINCH=6         " $B STK_X2W_WGS=384 STK_KSPLIT_MINCH=6
run "SS=0 G=3 WGS=384 KSWGS=768       " $B STK_X2W_WGS=384 STK_KSPLIT_WGS=768
run "SS=0 G=3 WGS=384 WP_SIDE=0       " $B STK_X2W_WGS=384 STK_WP_SIDE=0
run "SS=0 G=3 WGS=384 DY_PLANES=1     " $B STK_X2W_WGS=384 STK_DY_PLANES=1 STK_GN_BWD_PL_IPT=2
run "default                          " A=1
for W in celeba64 celebahq256; do
  for cfg in "A=1" "$B STK_X2W_WGS=384" "$B STK_X2W_WGS=512"; do
    ms=$(env $cfg python bench.py --workload $W --steps 12 --warmup 4 --no-cpu-baseline --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$W [$cfg]  $ms"
  done
done
