#!/bin/bash
# second in-situ sweep: the weight gradient's K split / slab traffic (see insitu_sweep.sh)
cd "$(dirname "$0")/../.."
run() {
  local label="$1"; shift
  local ms=$(env "$@" python bench.py --no-cpu-baseline --no-extra-workloads --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$label  $ms"
}
run "default                      " A=1
run "X2W_WGS=256                  " STK_X2W_WGS=256
run "X2W_WGS=320                  " STK_X2W_WGS=320
run "X2W_WGS=384                  " STK_X2W_WGS=384
run "X2W_WGS=448                  " STK_X2W_WGS=448
run "default                      " A=1
run "SLAB_MB=32                   " STK_WGRAD_SLAB_MB=32
run "SLAB_MB=48                   " STK_WGRAD_SLAB_MB=48
run "SLAB_MB=64                   " STK_WGRAD_SLAB_MB=64
run "SLAB_MB=96                   " STK_WGRAD_SLAB_MB=96
run "default                      " A=1
run "X2W_WGS=384 SLAB_MB=64       " STK_X2W_WGS=384 STK_WGRAD_SLAB_MB=64
run "X2W_WGS=384 SIDE_SHORTCUT=0  " STK_X2W_WGS=384 STK_SIDE_SHORTCUT=0
run "X2W_WGS=384 KSPLIT_MINCH=6   " STK_X2W_WGS=384 STK_KSPLIT_MINCH=6
run "X2W_WGS=384 W1_WGS=192       " STK_X2W_WGS=384 STK_W1_WGS=192
run "W1_WGS=128                   " STK_W1_WGS=128
run "W1_WGS=192                   " STK_W1_WGS=192
run "default                      " A=1
