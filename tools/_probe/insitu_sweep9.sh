#!/bin/bash
# under the round's final defaults: the main chain's own knobs once more
cd "$(dirname "$0")/../.."
run() {
  local label="$1"; shift
  local ms=$(timeout 150 env "$@" python bench.py --no-cpu-baseline --no-extra-workloads --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$label  $ms"
}
run "default                  " A=1
run "KSPLIT_WGS=256           " STK_KSPLIT_WGS=256
run "KSPLIT_WGS=384           " STK_KSPLIT_WGS=384
run "KSPLIT_WGS=768           " STK_KSPLIT_WGS=768
run "KSPLIT_MINCH=6           " STK_KSPLIT_MINCH=6
run "default                  " A=1
run "X2W_GROUPS=3 WGS=320     " STK_X2W_GROUPS=3 STK_X2W_WGS=320
run "X2W_GROUPS=3 WGS=256     " STK_X2W_GROUPS=3
run "WGRAD_SLAB_MB=32         " STK_WGRAD_SLAB_MB=32
run "GN_PL_T=1024             " STK_GN_PL_T=1024
run "X2D_HALO=2               " STK_X2D_HALO=2
run "REDUCE9=0                " STK_REDUCE9=0
run "default                  " A=1
