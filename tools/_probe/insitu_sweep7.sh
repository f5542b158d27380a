#!/bin/bash
# after the new defaults (STK_SIDE_SHORTCUT=0, STK_X2W_WGS=256): neighbours and the round's other switches again
cd "$(dirname "$0")/../.."
run() {
  local label="$1"; shift
  local ms=$(timeout 150 env "$@" python bench.py --no-cpu-baseline --no-extra-workloads --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$label  $ms"
}
run "default (new)          " A=1
run "old defaults           " STK_SIDE_SHORTCUT=1 STK_X2W_WGS=512
run "X2W_WGS=192            " STK_X2W_WGS=192
run "X2W_WGS=224            " STK_X2W_WGS=224
run "X2W_WGS=288            " STK_X2W_WGS=288
run "default (new)          " A=1
run "DY_PLANES=1 IPT=2      " STK_DY_PLANES=1 STK_GN_BWD_PL_IPT=2
run "DY_PLANES=1 IPT=4      " STK_DY_PLANES=1
run "X2D_T64=1              " STK_X2D_T64=1
run "WP_SIDE=0              " STK_WP_SIDE=0
run "SC_PEER_PLANES=0       " STK_SC_PEER_PLANES=0
run "KSPLIT_MINCH=6         " STK_KSPLIT_MINCH=6
run "SIDE_WGRAD1=0          " STK_SIDE_WGRAD1=0
run "W1_WGS=128             " STK_W1_WGS=128
run "X2W_GROUPS=1           " STK_X2W_GROUPS=1
run "default (new)          " A=1
