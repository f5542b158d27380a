#!/bin/bash
# Development: phase timings of the fused attention forward with parts of phase A removed (ATTN_EXPERIMENT builds).
# The variants are built through the library's own Makefile (so they carry its hazard flags and pass its post-link ISA scan)
# into a scratch copy of csrc/ -- the product library csrc/libstk.so is never touched; tools/attn_stamps.py loads the variant
# through STK_LIB (engine.lib.load_path).
set -e
SRC="$GRAFT_REPO_ROOT/soft-truncation_amd/csrc"
for E in 0 1 2 3; do
  OUT="/tmp/stk_attn_exp_$E"
  rm -rf "$OUT" && mkdir -p "$OUT/pkg/csrc" "$OUT/include" "$OUT/tools"
  cp "$SRC"/*.hip "$SRC"/*.h "$SRC"/Makefile "$OUT/pkg/csrc/"
  cp "$GRAFT_REPO_ROOT"/include/*.h "$OUT/include/"
  cp "$GRAFT_REPO_ROOT/tools/isa_scan.py" "$OUT/tools/"
  make -s -C "$OUT/pkg/csrc" -j4 CXXFLAGS="-O3 -std=c++17 -DATTN_EXPERIMENT=$E" INC="-I$OUT/include -I$OUT/pkg/csrc" ISA_SCAN="$OUT/tools/isa_scan.py"
  echo "== experiment $E"
  STK_LIB="$OUT/pkg/csrc/libstk.so" python "$GRAFT_REPO_ROOT/tools/attn_stamps.py" 2>&1 | tail -7
done
