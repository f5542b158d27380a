// common.h -- shared helpers for the gfx950 kernels of libstk (see include/stk.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stk.h"
#include "stk_rng.h"

#define STK_CHECK_LAUNCH()                          \
  do {                                              \
    if (hipGetLastError() != hipSuccess) return STK_ELAUNCH; \
  } while (0)

static inline int stk_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool stk_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// MI355X: 256 CUs.  Memory-bound grids are capped at 8 blocks of 256 threads per CU and
// grid-stride the rest (cdna_hip_programming.md Guideline 11).
constexpr int STK_NUM_CU = 256;
constexpr int STK_MAX_GRID = STK_NUM_CU * 8;
static inline int stk_ew_grid(long work_items, int threads = 256) {
  long g = (work_items + threads - 1) / threads;
  if (g < 1) g = 1;
  if (g > STK_MAX_GRID) g = STK_MAX_GRID;
  return (int)g;
}

// 64-lane wavefront reductions (wave = 64 on CDNA; never 32).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum of up to NV values per thread; `red` is LDS scratch of >= NV * (blockDim/64) floats.
// The result is broadcast to all threads.  Ends with a barrier so `red` can be reused immediately.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i * nw + wid] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[i * nw + w];
    v[i] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ float stk_sigmoid(float u) { return 1.f / (1.f + __expf(-u)); }

// XCD-aware bijective remap of a linear block id: the dispatcher places block b on XCD b % 8
// (speed assumption only), so give each XCD a contiguous range of logical tiles to keep
// neighbouring tiles -- which share operand panels -- in one L2 (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = id & 7, idx = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
