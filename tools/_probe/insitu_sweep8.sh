#!/bin/bash
cd "$(dirname "$0")/../.."
one() {
  local W="$1"; shift; local A="$1"; shift
  timeout 200 env "$@" python bench.py --workload $W $A --no-cpu-baseline --sampler-steps 0 --no-exchange-proxy --no-parity-probe --no-kernel-timer --prof-steps 0 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
}
for rep in 1 2 3 4; do
  echo "cifar10 r$rep [default]         $(one cifar10 '' A=1)"
  echo "cifar10 r$rep [DY_PLANES=1]     $(one cifar10 '' STK_DY_PLANES=1)"
  echo "cifar10 r$rep [DY_PLANES=1 IPT2] $(one cifar10 '' STK_DY_PLANES=1 STK_GN_BWD_PL_IPT=2)"
done
for W in celeba64 celebahq256; do for rep in 1 2; do
  echo "$W r$rep [default]       $(one $W '--steps 12 --warmup 4' A=1)"
  echo "$W r$rep [DY_PLANES=1]   $(one $W '--steps 12 --warmup 4' STK_DY_PLANES=1)"
done; done
