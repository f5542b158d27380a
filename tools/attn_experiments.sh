#!/bin/bash
# Development: phase timings of the fused attention forward with parts of phase A removed (ATTN_EXPERIMENT builds)
cd "$GRAFT_REPO_ROOT/soft-truncation_amd/csrc"
for E in 0 1 2 3; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DATTN_EXPERIMENT=$E -I../../include -I. -c attention.hip -o attention.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libstk.so elementwise.o upfirdn2d.o groupnorm.o reduce_optim.o attention.o conv.o
  echo "== experiment $E"
  python "$GRAFT_REPO_ROOT/tools/attn_stamps.py" 2>&1 | tail -7
done
