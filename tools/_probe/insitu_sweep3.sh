#!/bin/bash
cd "$(dirname "$0")/../.."
run() {
  local label="$1"; shift
  local ms=$(env "$@" python bench.py --no-cpu-baseline --no-extra-workloads --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$label  $ms"
}
for rep in 1 2; do
run "default                       " A=1
run "X2W_WGS=384                   " STK_X2W_WGS=384
run "SIDE_SHORTCUT=0               " STK_SIDE_SHORTCUT=0
run "X2W_WGS=384 SIDE_SHORTCUT=0   " STK_X2W_WGS=384 STK_SIDE_SHORTCUT=0
run "X2W_WGS=352 SIDE_SHORTCUT=0   " STK_X2W_WGS=352 STK_SIDE_SHORTCUT=0
run "X2W_WGS=416 SIDE_SHORTCUT=0   " STK_X2W_WGS=416 STK_SIDE_SHORTCUT=0
run "X2W_WGS=384 SS=0 SIDE_WGRAD1=0" STK_X2W_WGS=384 STK_SIDE_SHORTCUT=0 STK_SIDE_WGRAD1=0
run "X2W_WGS=384 SS=0 PEER_PLANES=0" STK_X2W_WGS=384 STK_SIDE_SHORTCUT=0 STK_SC_PEER_PLANES=0
run "X2W_WGS=384 SS=0 GROUPS=3     " STK_X2W_WGS=384 STK_SIDE_SHORTCUT=0 STK_X2W_GROUPS=3
done
