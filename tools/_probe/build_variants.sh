#!/bin/bash
# Builds variants of libstk.so for the co-residency hunt (tools/_probe/side_race2.py) into tools/_probe/build/:
#   libstk_slp.so        every translation unit WITH the SLP vectoriser (packed-fp32 VALU code), as in round 3 before the flag
#   libstk_gnslp.so      only groupnorm.hip with SLP (the kernel whose result went wrong), the rest as shipped
#   libstk_restslp.so    everything but groupnorm.hip with SLP
# Extra per-variant flags for groupnorm.hip: GN_EXTRA="-DSTK_GN_PROBE=1" etc.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/soft-truncation_amd/csrc
OUT=$ROOT/tools/_probe/build
mkdir -p $OUT
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter -I$ROOT/include -I$SRC"
SRCS="elementwise upfirdn2d groupnorm reduce_optim attention conv"
for s in $SRCS; do
  [ -f $OUT/${s}_slp.o ] && [ $OUT/${s}_slp.o -nt $SRC/$s.hip ] || /opt/rocm/bin/hipcc $BASE -c $SRC/$s.hip -o $OUT/${s}_slp.o &
done
wait
for s in $SRCS; do [ -f $OUT/${s}_slp.o ] || { echo "build of $s failed"; exit 1; }; done
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$1 "${@:2}"; }
N=$SRC   # shipped objects (no SLP)
link libstk_slp.so $OUT/elementwise_slp.o $OUT/upfirdn2d_slp.o $OUT/groupnorm_slp.o $OUT/reduce_optim_slp.o $OUT/attention_slp.o $OUT/conv_slp.o
link libstk_gnslp.so $N/elementwise.o $N/upfirdn2d.o $OUT/groupnorm_slp.o $N/reduce_optim.o $N/attention.o $N/conv.o
link libstk_restslp.so $OUT/elementwise_slp.o $OUT/upfirdn2d_slp.o $N/groupnorm.o $OUT/reduce_optim_slp.o $OUT/attention_slp.o $OUT/conv_slp.o
ls -la $OUT/*.so
