#!/bin/bash
# the packed-fp32 operand selections RCCL's gfx950 kernels use, beside no neighbour / AccVGPR MFMAs / VGPR MFMAs; one affected form as the control
cd "$(dirname "$0")/build"
timeout 500 ./cores2 -v form_fma_hi_from_lo0 -v form_mul_hi_from_lo0 -v form_add_hi_from_lo0 -v form_add_lo_from_hi1 -a none -a acc -a mfma_v
