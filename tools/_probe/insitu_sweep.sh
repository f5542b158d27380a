#!/bin/bash
# round 5: the tuning knobs of earlier rounds were set from micro-benchmarks; lesson 1 of this round says a kernel that is not
# matrix-bound behaves differently inside the step.  One-box A/B of each knob on the full training step (ms per step, 50 steps).
cd "$(dirname "$0")/../.."
run() {  # label, env assignments...
  local label="$1"; shift
  local ms=$(env "$@" python bench.py --no-cpu-baseline --no-extra-workloads --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$label  $ms"
}
run "default                " A=1
run "KSPLIT_WGS=256         " STK_KSPLIT_WGS=256
run "KSPLIT_WGS=768         " STK_KSPLIT_WGS=768
run "KSPLIT_MINCH=6         " STK_KSPLIT_MINCH=6
run "KSPLIT_MINCH=18        " STK_KSPLIT_MINCH=18
run "default                " A=1
run "X2W_WGS=384            " STK_X2W_WGS=384
run "X2W_WGS=768            " STK_X2W_WGS=768
run "X2W_GROUPS=1           " STK_X2W_GROUPS=1
run "X2W_GROUPS=3           " STK_X2W_GROUPS=3
run "WGRAD_SLAB_MB=64       " STK_WGRAD_SLAB_MB=64
run "default                " A=1
run "GN_PL_2K=0             " STK_GN_PL_2K=0
run "X2D_HALO=0             " STK_X2D_HALO=0
run "PL_KERNEL=3            " STK_PL_KERNEL=3
run "W1_WGS=512             " STK_W1_WGS=512
run "SIDE_SHORTCUT=0        " STK_SIDE_SHORTCUT=0
run "SIDE_WGRAD1=0          " STK_SIDE_WGRAD1=0
run "BWD_GRAPH=1            " STK_BWD_GRAPH=1
run "REDUCE9=0              " STK_REDUCE9=0
run "default                " A=1
