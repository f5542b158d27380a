#!/bin/bash
# small-map K-split sweep (tools/bench_x2d.py on the 8x8 / 4x4 shapes): workgroups to fill x fewest chunks per workgroup
cd "$(dirname "$0")/../.."
for cfg in "512 6" "256 6" "384 6" "768 6" "1024 6" "512 4" "512 9" "512 12" "768 4" "1024 3" "256 18"; do
  set -- $cfg
  echo "== STK_KSPLIT_WGS=$1 STK_KSPLIT_MINCH=$2"
  STK_KSPLIT_WGS=$1 STK_KSPLIT_MINCH=$2 python tools/bench_x2d.py --only 256x8,512x8,256x4,512x4 --tag w$1m$2 2>/dev/null | grep -E "fwd|dgrad" | awk '{printf "%s %s %s %s %s%s %s us %s TF/s %s\n",$2,$3,$4,$5,$7,$8,$9,$11,$13}'
done
