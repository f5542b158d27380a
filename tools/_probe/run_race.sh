#!/bin/bash
# One gpurun call: the reproducer on each library variant (tools/_probe/build_variants.sh built them).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/race
B=tools/_probe/build
for v in "$@"; do
  case $v in
    ship) L=soft-truncation_amd/csrc/libstk.so;;
    *) L=$B/libstk_$v.so;;
  esac
  echo "=== $v ($L)"
  PROBE_LIB=$L PROBE_TAG=$v timeout 900 python tools/_probe/side_race2.py 2>&1 | tail -40
done > gpurun_out/race/log_$(date +%H%M%S).txt 2>&1
tail -100 gpurun_out/race/log_*.txt
