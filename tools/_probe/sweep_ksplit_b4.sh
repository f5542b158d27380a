#!/bin/bash
# K-split sweep on the small-batch shapes of the 256x256 net (BASELINE configs[4]: per-GPU batch 4; sampling batch 16):
# workgroups to fill x fewest chunks per workgroup (tools/bench_x2d.py, forward / data gradient / weight gradient)
cd "$(dirname "$0")/../.."
for B in 4 16; do
for cfg in "512 12" "512 6" "512 4" "512 3" "512 2" "1024 3" "1024 2" "256 4"; do
  set -- $cfg
  echo "== batch $B STK_KSPLIT_WGS=$1 STK_KSPLIT_MINCH=$2"
  STK_KSPLIT_WGS=$1 STK_KSPLIT_MINCH=$2 python tools/bench_x2d.py --batch $B --shapes 256x32x256,512x32x256,256x16x256,512x16x256,256x8x256,512x8x256,256x4x256 --tag b${B}w$1m$2 2>/dev/null | grep -E "fwd|dgrad" | awk '{printf "%s %s %s %s %s %s%s %s us %s TF/s\n",$2,$3,$4,$5,$6,$7,$8,$9,$11}'
done
done
