#!/bin/bash
# per-kernel stats of the 64x64 and 256x256 nets (default engine), as kept in profiles/rNN_{celeba64,celebahq256}_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for w in celeba64 celebahq256; do
  O=$R/gpurun_out/${w}_kernel_stats.txt
  echo "(default engine = weight gradients on the side stream: traced durations include the time a kernel shares the chip)" > $O
  $R/tools/kstats_cmd.sh -n 45 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-exchange-proxy --no-extra-workloads --no-parity-probe --sampler-steps 0 --no-kernel-timer --prof-steps 0 >> $O 2>&1
done
