#!/bin/bash
cd "$(dirname "$0")/build"
V=""
for f in form_add_plain form_add_lo_from_hi1 form_add_lo_from_hi0 form_add_hi_from_lo1 form_add_hi_from_lo0 form_mul_lo_from_hi1 form_fma_lo_from_hi2 form_fma_hi_from_lo2 form_add_lo_from_hi1_b28 form_add_both_from_hi1 form_add_swap1 form_mul_sgpr_bcast form_mul_sgpr_pair form_mul_sgpr_src1_bcast; do V="$V -v $f"; done
timeout 500 ./cores2 $V -a none -a acc -a mfma_v
