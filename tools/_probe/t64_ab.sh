#!/bin/bash
# round 5: the small-tile kernel (STK_X2D_T64=1) against the K-split 128 x 128 form (=0) on the small-map shapes, batch 128 / 16 / 4
cd "$(dirname "$0")/../.."
for B in 128 16 4; do
  for T in 0 1; do
    for WG in 256 512; do
      [ $T = 0 ] && [ $WG = 512 ] && continue
      echo "== batch $B STK_X2D_T64=$T STK_T64_WGS=$WG"
      STK_X2D_T64=$T STK_T64_WGS=$WG python tools/bench_x2d.py --batch $B --shapes 256x32x256,256x16x256,512x16x256,256x8x256,512x8x256,256x4x256,512x4x256 --tag b${B}t$T$WG 2>/dev/null | grep -E "fwd|dgrad" | awk '{printf "%s %s %s %s %s %s%s %s us %s TF/s %s\n",$2,$3,$4,$5,$6,$7,$8,$9,$11,$13}'
    done
  done
done
