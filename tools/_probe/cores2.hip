// cores2.hip -- minimal reproducer of the round-3 co-residency divergence (see cores.hip for the full-kernel version).
//
// Finding of cores.hip: gn_bwd_flat_kernel<1, true> built WITH packed-fp32 code returns dx with the "- m1" term missing in
// lanes 48..63 of ONE packed result (the lo half of a v_pk_add_f32 whose lo result takes the HI register of a source pair,
// op_sel:[0,1]) whenever a wave that uses accumulation registers (AccVGPRs: v_accvgpr_* / MFMA on a[...]) is resident
// beside it.  Here: micro-victims made of the exact instruction window (inline asm, fixed registers) or of the single
// instruction, each checking its own results against scalar arithmetic in a loop, and micro-aggressors that isolate what
// the neighbour does.  Build with -fno-slp-vectorize (the checks must stay scalar).
//
//   cores2 [-v victim]... [-a aggressor]...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

// ---- victims: bad[lane * 4 + comp] += results that differ from the scalar evaluation; bad[256 + lane*4 + comp] += those that equal the
// evaluation with the "- m1" term dropped -----------------------------------------------------------------------------
struct Vals { float e[4], d[4]; };
__device__ __forceinline__ Vals expect(float gam, float invl, float dux, float duy, float duz, float duw, float g0, float g1,
                                       float xh0, float xh1, float xh2, float xh3, float rstd) {
  const float m1 = invl * g0, m2 = invl * g1, wz = gam * duz, ww = gam * duw;
  Vals v;
  v.e[0] = rstd * __fmaf_rn(-m2, xh0, __fmaf_rn(gam, dux, -m1));
  v.e[1] = rstd * __fmaf_rn(-m2, xh1, __fmaf_rn(gam, duy, -m1));
  v.e[2] = rstd * __fmaf_rn(-m2, xh2, wz - m1);
  v.e[3] = rstd * __fmaf_rn(-m2, xh3, ww - m1);
  v.d[0] = rstd * __fmaf_rn(-m2, xh0, gam * dux);
  v.d[1] = rstd * __fmaf_rn(-m2, xh1, gam * duy);
  v.d[2] = rstd * __fmaf_rn(-m2, xh2, wz);
  v.d[3] = rstd * __fmaf_rn(-m2, xh3, ww);
  return v;
}

// window: instructions (1)..(10) of the dx computation of gn_bwd_flat_kernel<1, true> as hipcc -O3 emits it with SLP on
template <int MODE>
__global__ __launch_bounds__(64) void vic_window(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, float* out) {
  const int l = threadIdx.x;
  const float* p = in + ((long)(blockIdx.x & 1023) * 64 + l) * 16;
  const float gam = p[0], invl = 1.f / 256.f, dux = p[1], duy = p[2], duz = p[3], duw = p[4], g0 = p[5], g1 = p[6];
  const float xh0 = p[7], xh1 = p[8], xh2 = p[9], xh3 = p[10], rstd = 3.5f;
  const unsigned long long rs = (unsigned long long)__float_as_uint(rstd) | 0x7fc0000000000000ULL;   // s[n] = rstd, s[n+1] = NaN (never read)
  const Vals ex = expect(gam, invl, dux, duy, duz, duw, g0, g1, xh0, xh1, xh2, xh3, rstd);
  unsigned b[4] = {0, 0, 0, 0}, dd[4] = {0, 0, 0, 0};
  float r0, r1, r2, r3;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      asm volatile(
        "v_mov_b32 v18, %4\n\tv_mov_b32 v19, %5\n\tv_mov_b32 v20, %6\n\tv_mov_b32 v21, %7\n\tv_mov_b32 v24, %8\n\tv_mov_b32 v26, %9\n\t"
        "v_mov_b32 v27, %10\n\tv_mov_b32 v29, %11\n\tv_mov_b32 v6, %12\n\tv_mov_b32 v7, %13\n\tv_mov_b32 v8, %14\n\tv_mov_b32 v9, %15\n\t"
        "s_nop 4\n\t"
        "v_pk_mul_f32 v[26:27], v[18:19], v[26:27]\n\t"
        "v_mul_f32_e32 v4, v19, v29\n\t"
        "v_mul_f32_e32 v32, v18, v24\n\t"
        "v_pk_fma_f32 v[2:3], v[18:19], v[20:21], v[26:27] op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
        "v_mov_b32_e32 v33, v26\n\t"
        "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 v[2:3], %16, v[2:3] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 v[4:5], v[4:5], v[8:9], v[6:7] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_mul_f32 v[4:5], %16, v[4:5] op_sel_hi:[0,1]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %0, v2\n\tv_mov_b32 %1, v3\n\tv_mov_b32 %2, v4\n\tv_mov_b32 %3, v5\n\t"
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
        : "v"(gam), "v"(invl), "v"(dux), "v"(duy), "v"(duz), "v"(duw), "v"(g0), "v"(g1), "v"(xh0), "v"(xh1), "v"(xh2), "v"(xh3), "s"(rs)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v18", "v19", "v20", "v21", "v24", "v26", "v27", "v29", "v32", "v33");
    } else {
      // the same arithmetic with s_nop 1 after every instruction (no back-to-back dependent issue)
      asm volatile(
        "v_mov_b32 v18, %4\n\tv_mov_b32 v19, %5\n\tv_mov_b32 v20, %6\n\tv_mov_b32 v21, %7\n\tv_mov_b32 v24, %8\n\tv_mov_b32 v26, %9\n\t"
        "v_mov_b32 v27, %10\n\tv_mov_b32 v29, %11\n\tv_mov_b32 v6, %12\n\tv_mov_b32 v7, %13\n\tv_mov_b32 v8, %14\n\tv_mov_b32 v9, %15\n\t"
        "s_nop 4\n\t"
        "v_pk_mul_f32 v[26:27], v[18:19], v[26:27]\n\ts_nop 4\n\t"
        "v_mul_f32_e32 v4, v19, v29\n\ts_nop 4\n\t"
        "v_mul_f32_e32 v32, v18, v24\n\ts_nop 4\n\t"
        "v_pk_fma_f32 v[2:3], v[18:19], v[20:21], v[26:27] op_sel:[0,0,1] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\ts_nop 4\n\t"
        "v_mov_b32_e32 v33, v26\n\ts_nop 4\n\t"
        "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\ts_nop 4\n\t"
        "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 4\n\t"
        "v_pk_mul_f32 v[2:3], %16, v[2:3] op_sel_hi:[0,1]\n\ts_nop 4\n\t"
        "v_pk_fma_f32 v[4:5], v[4:5], v[8:9], v[6:7] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\ts_nop 4\n\t"
        "v_pk_mul_f32 v[4:5], %16, v[4:5] op_sel_hi:[0,1]\n\t"
        "s_nop 4\n\t"
        "v_mov_b32 %0, v2\n\tv_mov_b32 %1, v3\n\tv_mov_b32 %2, v4\n\tv_mov_b32 %3, v5\n\t"
        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
        : "v"(gam), "v"(invl), "v"(dux), "v"(duy), "v"(duz), "v"(duw), "v"(g0), "v"(g1), "v"(xh0), "v"(xh1), "v"(xh2), "v"(xh3), "s"(rs)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v18", "v19", "v20", "v21", "v24", "v26", "v27", "v29", "v32", "v33");
    }
    const float r[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int j = 0; j < 4; ++j) { b[j] += r[j] != ex.e[j]; dd[j] += (r[j] != ex.e[j]) && (r[j] == ex.d[j]); }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (b[j]) atomicAdd(&bad[l * 4 + j], b[j]);
    if (dd[j]) atomicAdd(&bad[256 + l * 4 + j], dd[j]);
  }
  if (r0 == 1234.5f) out[0] = r0;
}

// the single instruction: d = (a.lo - b.hi, a.hi - b.hi); comp 0 = lo, comp 1 = hi
template <int MODE>
__global__ __launch_bounds__(64) void vic_single(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, float* out) {
  const int l = threadIdx.x;
  const float* p = in + ((long)(blockIdx.x & 1023) * 64 + l) * 16;
  const float a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3];
  const float e0 = MODE == 2 ? a0 - b0 : a0 - b1, e1 = a1 - b1;
  unsigned c0 = 0, c1 = 0, z0 = 0;
  float r0 = 0, r1 = 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0)          // sources and destination live in registers written long ago
      asm volatile("v_mov_b32 v32, %2\n\tv_mov_b32 v33, %3\n\tv_mov_b32 v26, %4\n\tv_mov_b32 v27, %5\n\ts_nop 7\n\t"
                   "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v6\n\tv_mov_b32 %1, v7"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v6", "v7", "v26", "v27", "v32", "v33");
    else if (MODE == 1)     // b.hi written by the instruction before (forwarding path)
      asm volatile("v_mov_b32 v32, %2\n\tv_mov_b32 v33, %3\n\tv_mov_b32 v26, %4\n\ts_nop 7\n\tv_mov_b32 v27, %5\n\t"
                   "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                   "v_mov_b32 %0, v6\n\tv_mov_b32 %1, v7"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v6", "v7", "v26", "v27", "v32", "v33");
    else                    // no swizzle: d = (a.lo - b.lo, a.hi - b.hi)
      asm volatile("v_mov_b32 v32, %2\n\tv_mov_b32 v33, %3\n\tv_mov_b32 v26, %4\n\tv_mov_b32 v27, %5\n\ts_nop 7\n\t"
                   "v_pk_add_f32 v[6:7], v[32:33], v[26:27] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 7\n\t"
                   "v_mov_b32 %0, v6\n\tv_mov_b32 %1, v7"
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v6", "v7", "v26", "v27", "v32", "v33");
    c0 += r0 != e0; c1 += r1 != e1; z0 += (r0 != e0) && (r0 == a0);
  }
  if (c0) atomicAdd(&bad[l * 4 + 0], c0);
  if (c1) atomicAdd(&bad[l * 4 + 1], c1);
  if (z0) atomicAdd(&bad[256 + l * 4 + 0], z0);
  if (r0 == 1234.5f) out[0] = r0 + r1;
}

// instruction forms: which operand selections are affected.  bad[lane*4 + {0: lo, 1: hi}], bad[256 + ...] = wrong AND equal to the
// result with the swizzled operand read as +0
#define FORM_KERNEL(NAME, ASM, E0, E1, Z0, Z1)                                                                            \
  __global__ __launch_bounds__(64) void NAME(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, float* out) { \
    const int l = threadIdx.x;                                                                                            \
    const float* p = in + ((long)(blockIdx.x & 1023) * 64 + l) * 16;                                                      \
    const float a0 = p[0], a1 = p[1], b0 = p[2], b1 = p[3], c0 = p[4], c1 = p[5];                                         \
    const float s0 = in[7], s1 = in[8];                                                                                   \
    const unsigned long long sp = (unsigned long long)__float_as_uint(s0) | ((unsigned long long)__float_as_uint(s1) << 32); \
    const float e0 = (E0), e1 = (E1), z0 = (Z0), z1 = (Z1);                                                               \
    unsigned k0 = 0, k1 = 0, y0 = 0, y1 = 0;                                                                              \
    float r0 = 0, r1 = 0;                                                                                                 \
    for (int i = 0; i < iters; ++i) {                                                                                     \
      asm volatile("v_mov_b32 v32, %2\n\tv_mov_b32 v33, %3\n\tv_mov_b32 v26, %4\n\tv_mov_b32 v27, %5\n\tv_mov_b32 v28, %4\n\tv_mov_b32 v29, %5\n\t" \
                   "v_mov_b32 v20, %6\n\tv_mov_b32 v21, %7\n\ts_nop 7\n\t" ASM "\n\ts_nop 7\n\tv_mov_b32 %0, v6\n\tv_mov_b32 %1, v7"           \
                   : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "s"(sp)                    \
                   : "v6", "v7", "v20", "v21", "v26", "v27", "v28", "v29", "v32", "v33");                                  \
      k0 += r0 != e0; k1 += r1 != e1; y0 += (r0 != e0) && (r0 == z0); y1 += (r1 != e1) && (r1 == z1);                     \
    }                                                                                                                     \
    if (k0) atomicAdd(&bad[l * 4 + 0], k0);                                                                               \
    if (k1) atomicAdd(&bad[l * 4 + 1], k1);                                                                               \
    if (y0) atomicAdd(&bad[256 + l * 4 + 0], y0);                                                                         \
    if (y1) atomicAdd(&bad[256 + l * 4 + 1], y1);                                                                         \
    if (r0 == 1234.5f) out[0] = r0 + r1;                                                                                  \
  }
FORM_KERNEL(form_add_lo_from_hi1, "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel:[0,1]", a0 + b1, a1 + b1, a0, a1)
FORM_KERNEL(form_add_lo_from_hi0, "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel:[1,0]", a1 + b0, a1 + b1, b0, b1)
FORM_KERNEL(form_add_hi_from_lo1, "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel_hi:[1,0]", a0 + b0, a1 + b0, a0, a1)
FORM_KERNEL(form_add_hi_from_lo0, "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel_hi:[0,1]", a0 + b0, a0 + b1, b0, b1)
FORM_KERNEL(form_mul_lo_from_hi1, "v_pk_mul_f32 v[6:7], v[32:33], v[26:27] op_sel:[0,1]", a0 * b1, a1 * b1, a0 * 0.f, a1 * 0.f)
FORM_KERNEL(form_fma_lo_from_hi2, "v_pk_fma_f32 v[6:7], v[32:33], v[26:27], v[20:21] op_sel:[0,0,1] op_sel_hi:[1,1,1]", __fmaf_rn(a0, b0, c1), __fmaf_rn(a1, b1, c1), a0 * b0, a1 * b1)
FORM_KERNEL(form_fma_hi_from_lo2, "v_pk_fma_f32 v[6:7], v[32:33], v[26:27], v[20:21] op_sel_hi:[1,1,0]", __fmaf_rn(a0, b0, c0), __fmaf_rn(a1, b1, c0), a0 * b0, a1 * b1)
FORM_KERNEL(form_add_plain, "v_pk_add_f32 v[6:7], v[32:33], v[26:27]", a0 + b0, a1 + b1, a0, a1)
FORM_KERNEL(form_add_lo_from_hi1_b28, "v_pk_add_f32 v[6:7], v[32:33], v[28:29] op_sel:[0,1]", a0 + b1, a1 + b1, a0, a1)
FORM_KERNEL(form_mul_sgpr_bcast, "v_pk_mul_f32 v[6:7], %8, v[26:27] op_sel_hi:[0,1]", s0 * b0, s0 * b1, 0.f * b0, 0.f * b1)
FORM_KERNEL(form_mul_sgpr_pair, "v_pk_mul_f32 v[6:7], %8, v[26:27]", s0 * b0, s1 * b1, 0.f * b0, 0.f * b1)
FORM_KERNEL(form_mul_sgpr_src1_bcast, "v_pk_mul_f32 v[6:7], v[26:27], %8 op_sel_hi:[1,0]", s0 * b0, s0 * b1, 0.f * b0, 0.f * b1)
FORM_KERNEL(form_fma_lo_from_hi1, "v_pk_fma_f32 v[6:7], v[32:33], v[26:27], v[20:21] op_sel:[0,1,0] op_sel_hi:[1,1,1]", __fmaf_rn(a0, b1, c0), __fmaf_rn(a1, b1, c1), __fmaf_rn(a0, 0.f, c0), c1)
FORM_KERNEL(form_fma_lo_from_hi0, "v_pk_fma_f32 v[6:7], v[32:33], v[26:27], v[20:21] op_sel:[1,0,0] op_sel_hi:[1,1,1]", __fmaf_rn(a1, b0, c0), __fmaf_rn(a1, b1, c1), __fmaf_rn(0.f, b0, c0), c1)
FORM_KERNEL(form_mul_lo_from_hi0, "v_pk_mul_f32 v[6:7], v[32:33], v[26:27] op_sel:[1,0]", a1 * b0, a1 * b1, 0.f * b0, 0.f)
FORM_KERNEL(form_mov_lo_from_hi, "v_pk_mov_b32 v[6:7], v[32:33], v[26:27] op_sel:[1,1]", a1, b1, 0.f, 0.f)
// the forms found in RCCL's gfx950 kernels (src0 broadcast into the HI result)
FORM_KERNEL(form_fma_hi_from_lo0, "v_pk_fma_f32 v[6:7], v[32:33], v[26:27], v[20:21] op_sel_hi:[0,1,1]", __fmaf_rn(a0, b0, c0), __fmaf_rn(a0, b1, c1), __fmaf_rn(a0, 0.f, c0), __fmaf_rn(a0, 0.f, c1))
FORM_KERNEL(form_mul_hi_from_lo0, "v_pk_mul_f32 v[6:7], v[32:33], v[26:27] op_sel_hi:[0,1]", a0 * b0, a0 * b1, a0 * 0.f, a0 * 0.f)
FORM_KERNEL(form_add_both_from_hi1, "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel:[0,1] op_sel_hi:[1,1]", a0 + b1, a1 + b1, a0, a1)
FORM_KERNEL(form_add_swap1, "v_pk_add_f32 v[6:7], v[32:33], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]", a0 + b1, a1 + b0, a0, a1)

// ---- aggressors -------------------------------------------------------------------------------------------------
#define ACLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
__global__ __launch_bounds__(256) void agg_acc(float* out, int iters) {        // the trigger found by cores.hip
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i)
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a100, %0\n\ts_nop 4\n\t"
                 "v_mfma_f32_32x32x16_f16 a[0:15], %1, %2, a[0:15]\n\ts_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a5\n\tv_accvgpr_mov_b32 a101, a100"
                 : "+v"(v) : "v"(a), "v"(b) : ACLOB, "a100", "a101");
  if (v == 12345.678f) out[0] = v;
}
__global__ __launch_bounds__(256) void agg_acc_w(float* out, int iters) {      // only writes to accumulation registers
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i)
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0" : : "v"(v) : ACLOB);
  if (v == 12345.678f) out[0] = v;
}
__global__ __launch_bounds__(256) void agg_acc_r(float* out, int iters) {      // only reads of accumulation registers
  float v = threadIdx.x, w = 0;
  asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\ts_nop 4" : : "v"(v) : ACLOB);
  for (int i = 0; i < iters; ++i)
    asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %0, a1\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %0, a1" : "=v"(w) : : ACLOB);
  if (w == 12345.678f) out[0] = w;
}
__global__ __launch_bounds__(256) void agg_acc_mfma(float* out, int iters) {   // MFMA accumulating in a[0:15], nothing else
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float v = 0;
  asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\ts_nop 4" : : "v"(v) : ACLOB);
  for (int i = 0; i < iters; ++i)
    asm volatile("v_mfma_f32_32x32x16_f16 a[0:15], %0, %1, a[0:15]" : : "v"(a), "v"(b) : ACLOB);
  asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) : : ACLOB);
  if (v == 12345.678f) out[0] = v;
}
__global__ __launch_bounds__(256) void agg_acc_alloc(float* out, int iters) {  // accumulation registers allocated, never touched in the loop
  float v = threadIdx.x;
  asm volatile("v_accvgpr_write_b32 a100, %0" : : "v"(v) : ACLOB, "a100");
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
  if (v == 12345.678f) out[0] = v;
}
__global__ __launch_bounds__(256) void agg_mfma_v(float* out, int iters) {     // MFMA accumulating in ordinary VGPRs (control)
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float16v c = {0};
  for (int i = 0; i < iters; ++i)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  asm volatile("s_nop 15\n\ts_nop 15");
  if (c[0] == 12345.678f) out[0] = c[3];
}

int main(int argc, char** argv) {
  std::vector<std::string> vics, aggs;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "-v")) vics.push_back(argv[i + 1]);
    else if (!strcmp(argv[i], "-a")) aggs.push_back(argv[i + 1]);
  }
  if (vics.empty()) vics = {"window", "window_nop", "single", "single_fwd", "single_plain"};
  if (aggs.empty()) aggs = {"none", "mfma_v", "acc", "acc_w", "acc_r", "acc_mfma", "acc_alloc"};
  const int VB = 2304, iters = getenv("CORES_VITERS") ? atoi(getenv("CORES_VITERS")) : 400;
  std::mt19937 rng(99);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hin(1024L * 64 * 16);
  for (auto& v : hin) v = nd(rng);
  float *d_in, *d_out; unsigned* d_bad;
  CK(hipMalloc(&d_in, hin.size() * 4)); CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_bad, 512 * 4));
  CK(hipMemcpy(d_in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(d_out, 0, 4096));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int asc = getenv("CORES_ITERS") ? atoi(getenv("CORES_ITERS")) : 1;
  for (const auto& vc : vics)
    for (const auto& ag : aggs) {
      CK(hipMemset(d_bad, 0, 512 * 4));
      if (ag == "acc") hipLaunchKernelGGL(agg_acc, dim3(512), dim3(256), 0, sb, d_out, 300000 * asc);
      else if (ag == "acc_w") hipLaunchKernelGGL(agg_acc_w, dim3(512), dim3(256), 0, sb, d_out, 6000000 * asc);
      else if (ag == "acc_r") hipLaunchKernelGGL(agg_acc_r, dim3(512), dim3(256), 0, sb, d_out, 6000000 * asc);
      else if (ag == "acc_mfma") hipLaunchKernelGGL(agg_acc_mfma, dim3(512), dim3(256), 0, sb, d_out, 1000000 * asc);
      else if (ag == "acc_alloc") hipLaunchKernelGGL(agg_acc_alloc, dim3(512), dim3(256), 0, sb, d_out, 12000 * asc);
      else if (ag == "mfma_v") hipLaunchKernelGGL(agg_mfma_v, dim3(512), dim3(256), 0, sb, d_out, 1000000 * asc);
      else if (ag != "none") { fprintf(stderr, "unknown aggressor %s\n", ag.c_str()); return 2; }
      CK(hipGetLastError());
      for (int k = 0; k < 8; ++k) {
        if (vc == "window") hipLaunchKernelGGL(vic_window<0>, dim3(VB), dim3(64), 0, sa, d_in, d_bad, iters, d_out);
        else if (vc == "window_nop") hipLaunchKernelGGL(vic_window<1>, dim3(VB), dim3(64), 0, sa, d_in, d_bad, iters, d_out);
        else if (vc == "single") hipLaunchKernelGGL(vic_single<0>, dim3(VB), dim3(64), 0, sa, d_in, d_bad, iters, d_out);
        else if (vc == "single_fwd") hipLaunchKernelGGL(vic_single<1>, dim3(VB), dim3(64), 0, sa, d_in, d_bad, iters, d_out);
        else if (vc == "single_plain") hipLaunchKernelGGL(vic_single<2>, dim3(VB), dim3(64), 0, sa, d_in, d_bad, iters, d_out);
#define FORM_CASE(NAME) else if (vc == #NAME) hipLaunchKernelGGL(NAME, dim3(VB), dim3(64), 0, sa, d_in, d_bad, iters, d_out);
        FORM_CASE(form_add_lo_from_hi1) FORM_CASE(form_add_lo_from_hi0) FORM_CASE(form_add_hi_from_lo1) FORM_CASE(form_add_hi_from_lo0)
        FORM_CASE(form_mul_lo_from_hi1) FORM_CASE(form_fma_lo_from_hi2) FORM_CASE(form_fma_hi_from_lo2) FORM_CASE(form_add_plain)
        FORM_CASE(form_add_lo_from_hi1_b28) FORM_CASE(form_mul_sgpr_bcast) FORM_CASE(form_mul_sgpr_pair) FORM_CASE(form_mul_sgpr_src1_bcast)
        FORM_CASE(form_add_both_from_hi1) FORM_CASE(form_add_swap1)
        FORM_CASE(form_fma_hi_from_lo0) FORM_CASE(form_mul_hi_from_lo0)
        FORM_CASE(form_fma_lo_from_hi1) FORM_CASE(form_fma_lo_from_hi0) FORM_CASE(form_mul_lo_from_hi0) FORM_CASE(form_mov_lo_from_hi)
        else { fprintf(stderr, "unknown victim %s\n", vc.c_str()); return 2; }
      }
      CK(hipGetLastError());
      CK(hipStreamSynchronize(sa));
      const bool busy = hipStreamQuery(sb) == hipErrorNotReady;
      CK(hipStreamSynchronize(sb));
      unsigned hb[512];
      CK(hipMemcpy(hb, d_bad, sizeof hb, hipMemcpyDeviceToHost));
      unsigned long long tot = 0, dropped = 0;
      for (int i = 0; i < 256; ++i) { tot += hb[i]; dropped += hb[256 + i]; }
      printf("victim %-12s aggressor %-9s (%s) wrong results %llu of %llu", vc.c_str(), ag.c_str(), busy ? "resident throughout" : "ENDED EARLY", tot,
             8ULL * VB * 64 * iters * (vc.rfind("window", 0) == 0 ? 4 : 2));
      if (tot) {
        printf(" (= term dropped: %llu); lanes:", dropped);
        for (int l = 0; l < 64; ++l) if (hb[l * 4] | hb[l * 4 + 1] | hb[l * 4 + 2] | hb[l * 4 + 3]) printf(" %d", l);
        printf("; components:");
        for (int c = 0; c < 4; ++c) { unsigned long long s = 0; for (int l = 0; l < 64; ++l) s += hb[l * 4 + c]; if (s) printf(" %d(%llu)", c, s); }
      }
      printf("\n");
      fflush(stdout);
    }
  return 0;
}
