// attention.hip -- the attention core of AttnBlockpp (models/layerspp.py:95-99) as fused kernels for gfx950:
//
//     s[t, t'] = C^-0.5 sum_c q[c, t] k[c, t']        p = softmax over t'        o[c, t] = sum_t' v[c, t'] p[t, t']
//
// on q, k, v, o of shape [B, C, T] (NCHW with T = H W <= 256 positions, one head of all C <= 256 channels), forward and
// backward, without a [B, T, T] matrix in HBM.  The reference runs two einsums and a softmax (five more einsums and a
// softmax backward under autograd); round 1 ran them as batched GEMMs on the f32-input MFMA (157 TFLOP/s peak) with the
// score and probability matrices written to and read from HBM.
//
// Arithmetic.  Every product is an fp32 product evaluated on the fp16 matrix pipe from two-way split operands, as in the
// convolutions (conv_x2.h): an operand tensor is multiplied by the power of two that puts its largest magnitude in
// [2^13, 2^14), each value becomes hi + lo fp16 terms (11 + 11 bits), a product is hi hi + (hi lo + lo hi) -- three
// v_mfma_f32_32x32x16_f16 with fp32 accumulation, error O(2^-22).  The scales of q, k, v, do come from |x| passes (256
// partial maxima per tensor, reduced by every workgroup: no finishing launch); probabilities are <= 1 (fixed scale
// 2^13); the score gradient ds is scaled by its workgroup's own maximum, which is exact because a workgroup holds whole
// rows (columns) of ds, so the scale factors out of its contraction.  softmax uses the accurate expf / logf.
//
// One kernel template, three modes.  A workgroup (256 threads, one per CU: it uses up to 150 KB of LDS) owns a tile of 64
// positions on the "n" side and all T positions on the "m" side:
//   FWD  n = queries t.   phase A: D[m = t', n = t] = sum_c k[c, t'] q[c, t];  softmax over m in registers (+ LDS across
//        the four waves);  phase B: o[c, t] = sum_t' v[c, t'] p~[t, t'] / rowsum[t];  writes lse[t] for the backward.
//   DQ   n = queries t.   phase A: S^T and dP^T[m = t', n = t] = sum_c v[c, t'] do[c, t] together;  p = exp(s - lse),
//        delta[t] = sum_t' p dp (written out for DKV),  ds = scale p (dp - delta);  phase B: dq[c, t] = sum_t' k[c, t'] ds[t, t'].
//   DKV  n = keys t'.     phase A: S and dP[m = t, n = t'];  p, ds from lse[t], delta[t];  phase B twice:
//        dv[c, t'] = sum_t do[c, t] p[t, t'],  dk[c, t'] = sum_t q[c, t] ds[t, t'].
// The backward recomputes the scores (7 GEMM units instead of 5) and never exchanges data between workgroups, so it is
// deterministic.
//
// Phase A contracts over channels, the SLOW index of both operands in NCHW: tiles are staged as [32-position block]
// [channel][32 positions] fp16 (64-byte rows, conflict-free) and the MFMA fragments come from `ds_read_b64_tr_b16`, which
// hands every lane 4 consecutive channels of its position (semantics pinned on the hardware by tools/_probe/tr.hip).
// Phase B contracts over positions, the fast index of its A operand (v, k, do, q rows) and of the probabilities written
// back from the accumulators as [n][m]: plain `ds_read_b128` fragments from padded rows (80 / 528-byte pitch).  Both
// phases are software-pipelined: global fp32 loads of chunk c + 2 in flight, chunk c + 1 being split and written to the
// other LDS buffer, MFMAs on chunk c, one barrier per chunk.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((__vector_size__(8)));
typedef __attribute__((address_space(3))) s4 lds_s4;

constexpr int NPART = 256;          // partial |x| maxima per tensor (a "scale record", include/stk.h)
constexpr int MODE_FWD = 0, MODE_DQ = 1, MODE_DKV = 2;

// ---- |x| maxima of up to four tensors of n floats each in one launch: blockIdx.y picks the tensor ----------------------
struct AmaxArgs { const float* x[4]; float* rec[4]; long n; };
__global__ __launch_bounds__(1024) void attn_amax_kernel(AmaxArgs a) {
  __shared__ float red[16];
  const float* x = a.x[blockIdx.y];
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const long n4 = a.n >> 2;                                   // n % 4 == 0, 16-byte aligned (checked by the host)
  float m0 = 0.f, m1 = 0.f;
  const long stride = (long)NPART * 1024;
  long i = (long)blockIdx.x * 1024 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 u = x4[i], v = x4[i + stride];
    m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
    m1 = fmaxf(fmaxf(m1, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (i < n4) {
    const float4 u = x4[i];
    m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
  }
  float m = wave_max(fmaxf(m0, m1));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < 16; ++w) m = fmaxf(m, red[w]);
    a.rec[blockIdx.y][blockIdx.x] = m;
  }
}

// power of two s with m s in [2^13, 2^14) (1 for m = 0) -- the same rule as x2::pow2_scale_of (conv_x2.h)
__device__ __forceinline__ float pow2_scale_of(float m) {
  const int be = (int)((__float_as_uint(m) >> 23) & 0xffu);
  if (be == 0) return 1.f;
  const int se = min(max(127 + 13 - (be - 127), 1), 254);
  return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ unsigned pack_h2(float a, float b) {
  const halfx2 v = {(_Float16)a, (_Float16)b};                  // round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_part(float v) { return v - (float)(_Float16)v; }      // exact
// four scaled fp32 values -> hi and lo fp16 quadruples (8 bytes each)
__device__ __forceinline__ void split4(const float (&v)[4], u32x2& hi, u32x2& lo) {
  hi = u32x2{pack_h2(v[0], v[1]), pack_h2(v[2], v[3])};
  lo = u32x2{pack_h2(lo_part(v[0]), lo_part(v[1])), pack_h2(lo_part(v[2]), lo_part(v[3]))};
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ halfx8 cat8(s4 lo, s4 hi) {
  typedef short s8 __attribute__((__vector_size__(16)));
  const s8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(halfx8, v);
}
// row of accumulator element e of a 32x32 tile, relative to 4 * (lane / 32); the column is lane % 32
__device__ __forceinline__ constexpr int acc_row(int e) { return (e & 3) + 8 * (e >> 2); }

struct Args {
  const float* x[2];      // phase A operands on the m side (rows of D), [B, C, T]
  const float* y[2];      // phase A operands on the n side (columns of D)
  const float* a[2];      // phase B operands [B, C, T] (rows c, contraction along T)
  const float* rx[2]; const float* ry[2]; const float* ra[2];    // their scale records
  float* out[2]; float beta[2];                                  // phase B results [B, C, T]
  float* lse;             // [B, T]  FWD: written; DQ / DKV: read
  float* delta;           // [B, T]  DQ: written; DKV: read
  int B, C, T; float scale;
};

template <int MODE, int TK>
struct Geo {
  static constexpr int NG = MODE == MODE_FWD ? 1 : 2;     // GEMMs of phase A
  static constexpr int CK = 32 / NG;                      // channels per phase A chunk
  static constexpr int KS = CK / 16;                      // MFMA k steps per chunk
  static constexpr int MBW = TK / 128;                    // 32-row blocks of D per wave
  static constexpr int XBLK = TK / 32;
  static constexpr int XP = TK * CK * 2;                  // bytes per plane of an X chunk tile [m block][c][32]
  static constexpr int YP = 64 * CK * 2;
  static constexpr int GSZ = 2 * XP + 2 * YP;             // one GEMM's tiles: X hi, X lo, Y hi, Y lo
  static constexpr int STAGE = NG * GSZ;                  // 40960 (TK = 256) / 24576 (TK = 128)
  static constexpr int XI = XBLK * (CK / 8) / 4;          // 16-byte loads per thread, chunk and X operand
  static constexpr int YI = 2 * (CK / 8) / 4;             //                                 ... and Y operand (0 -> see YI1)
  static constexpr int YI1 = YI > 0 ? YI : 1;
  static constexpr int APITCH = 80;                       // phase B: A block [256 c][32 k] fp16, padded rows
  static constexpr int AP = 256 * APITCH;
  static constexpr int ASTAGE = 2 * AP;                   // 40960
  static constexpr int BPITCH = TK * 2 + 16;              // phase B: probabilities [64 n][TK k] fp16, padded rows
  static constexpr int BP = 64 * BPITCH;
  static constexpr int REGION0 = 2 * STAGE > 2 * ASTAGE ? 2 * STAGE : 2 * ASTAGE;     // 81920
  static constexpr int LDS = REGION0 + 2 * BP;
};

template <int MODE, int TK>
__global__ __launch_bounds__(256, 1) void attn_kernel(Args a) {
  using G = Geo<MODE, TK>;
  constexpr int NG = G::NG, CK = G::CK, KS = G::KS, MBW = G::MBW, XBLK = G::XBLK;
  static_assert(G::YI >= 1 || MODE != MODE_FWD, "geometry");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS];
  __shared__ float red[2][4][64];                 // cross-wave column statistics
  __shared__ float rowstat[2][TK];                // DKV: lse[m], delta[m]
  __shared__ float sred[8];
  unsigned char* const bm = lds + G::REGION0;     // probabilities / score gradients for phase B

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = (a.T + 63) >> 6;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int b = id / ntile, n0 = (id - b * ntile) * 64;
  const int T = a.T, C = a.C;

  // ---- scales of the tensor operands: maxima of their 256-entry records -------------------------------------------------
  float sx[2] = {1.f, 1.f}, sy[2] = {1.f, 1.f}, sa[2] = {1.f, 1.f};
  {
    float m[6];
    m[0] = a.rx[0][tid]; m[1] = a.ry[0][tid]; m[2] = a.ra[0][tid];
    m[3] = NG == 2 ? a.rx[1][tid] : 0.f; m[4] = NG == 2 ? a.ry[1][tid] : 0.f; m[5] = MODE == MODE_DKV ? a.ra[1][tid] : 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = wave_max(m[i]);
    float* r6 = reinterpret_cast<float*>(lds);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) r6[wid * 6 + i] = m[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = fmaxf(fmaxf(r6[i], r6[6 + i]), fmaxf(r6[12 + i], r6[18 + i]));
    __syncthreads();
    sx[0] = pow2_scale_of(m[0]); sy[0] = pow2_scale_of(m[1]); sa[0] = pow2_scale_of(m[2]);
    sx[1] = pow2_scale_of(m[3]); sy[1] = pow2_scale_of(m[4]); sa[1] = pow2_scale_of(m[5]);
  }
  if (MODE == MODE_DKV) {
    for (int m = tid; m < TK; m += 256) {
      rowstat[0][m] = m < T ? a.lse[(long)b * T + m] : INFINITY;       // exp(s - inf) = 0 for the padding rows
      rowstat[1][m] = m < T ? a.delta[(long)b * T + m] : 0.f;
    }
  }

  // ---- phase A: D_g[m, n] = sum_c X_g[c, m] Y_g[c, n] --------------------------------------------------------------------
  const long tensor_bytes = (long)a.B * C * T * 4;
  const int l8 = lane >> 3, l7 = lane & 7;
  __amdgpu_buffer_rsrc_t xrs[NG], yrs[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) { xrs[g] = make_rsrc(a.x[g], tensor_bytes); yrs[g] = make_rsrc(a.y[g], tensor_bytes); }
  // a 16-byte load covers 4 positions of one channel; a wave instruction 8 channels x 32 positions (one 32-block)
  unsigned xvo[G::XI], xdst[G::XI], yvo[G::YI1], ydst[G::YI1];
#pragma unroll
  for (int j = 0; j < G::XI; ++j) {
    const int i = wid + 4 * j, mblk = i % XBLK, c = (i / XBLK) * 8 + l8, m = mblk * 32 + 4 * l7;
    xvo[j] = (unsigned)(c * T + m) * 4u | (m < T ? 0u : 0x80000000u);          // bit 31: out of range -> zeros
    xdst[j] = (unsigned)((mblk * CK + c) * 64 + 8 * l7);
  }
  // Y tiles: 2 blocks x CK / 8 instructions; with CK = 16 that is one instruction per wave, with CK = 32 two
#pragma unroll
  for (int j = 0; j < G::YI1; ++j) {
    const int i = wid + 4 * j, nblk = i & 1, c = (i >> 1) * 8 + l8, n = nblk * 32 + 4 * l7;
    yvo[j] = (unsigned)(c * T + n0 + n) * 4u | (n0 + n < T ? 0u : 0x80000000u);
    ydst[j] = (unsigned)((nblk * CK + c) * 64 + 8 * l7);
  }
  u32x4 xr[NG][G::XI], yr[NG][G::YI1];
  auto fetch = [&](int chunk) __attribute__((always_inline)) {
    const unsigned soff = (unsigned)((b * C + chunk * CK) * T) * 4u;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int j = 0; j < G::XI; ++j)
        xr[g][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs[g], (int)xvo[j], (int)soff, 0));
#pragma unroll
      for (int j = 0; j < G::YI1; ++j)
        yr[g][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs[g], (int)yvo[j], (int)soff, 0));
    }
  };
  auto put = [&](unsigned char* hi_plane, int plane_bytes, unsigned dst, u32x4 r, float s) __attribute__((always_inline)) {
    const float v[4] = {s * __uint_as_float(r[0]), s * __uint_as_float(r[1]), s * __uint_as_float(r[2]), s * __uint_as_float(r[3])};
    u32x2 hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<u32x2*>(hi_plane + dst) = hi;
    *reinterpret_cast<u32x2*>(hi_plane + plane_bytes + dst) = lo;
  };
  auto stage = [&](unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int j = 0; j < G::XI; ++j) put(buf + g * G::GSZ, G::XP, xdst[j], xr[g][j], sx[g]);
#pragma unroll
      for (int j = 0; j < G::YI1; ++j) put(buf + g * G::GSZ + 2 * G::XP, G::YP, ydst[j], yr[g][j], sy[g]);
    }
  };

  floatx16 acc[NG][MBW][2];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[g][mb][nb][e] = 0.f;

  // transpose-read addresses: lane (mm = lane % 16, g4 = lane / 16) supplies the 8 bytes at channel k0 + mm / 4,
  // positions r0 + 4 (mm % 4) .. + 3 and receives channels k0 .. k0 + 3 of position r0 + mm:  r0 = 16 (g4 & 1),
  // k0 = 8 (g4 >> 1) + 4 h
  const int mm = lane & 15, g4 = lane >> 4;
  const int tr0 = (8 * (g4 >> 1) + (mm >> 2)) * 64 + (16 * (g4 & 1) + 4 * (mm & 3)) * 2;
  auto tr8 = [&](const unsigned char* p) __attribute__((always_inline)) {
    return cat8(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + tr0)),
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(p + tr0 + 4 * 64)));
  };
  constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};        // cross terms first (fixed accumulation order)
  auto mma_a = [&](const unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        halfx8 xa[MBW][2], yb[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb)
            xa[mb][p] = tr8(buf + g * G::GSZ + p * G::XP + ((wid * MBW + mb) * CK + kk * 16) * 64);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            yb[nb][p] = tr8(buf + g * G::GSZ + 2 * G::XP + p * G::YP + (nb * CK + kk * 16) * 64);
        }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[g][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[mb][SA[pr]], yb[nb][SB[pr]], acc[g][mb][nb], 0, 0, 0);
      }
  };

  const int nchunk = C / CK;
  fetch(0);
  stage(lds);
  if (nchunk > 1) fetch(1);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    unsigned char* cur = lds + (c & 1) * G::STAGE;
    if (c + 1 < nchunk) stage(lds + ((c + 1) & 1) * G::STAGE);
    if (c + 2 < nchunk) fetch(c + 2);
    mma_a(cur);
    __syncthreads();
  }

  // ---- phase B plumbing ----------------------------------------------------------------------------------------------
  __amdgpu_buffer_rsrc_t ars = make_rsrc(a.a[0], tensor_bytes);
  const int KB = (T + 31) >> 5;                               // 32-position blocks of the contraction
  const int cw = C >> 5;                                       // live 32-channel blocks
  unsigned avo[8], adst[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = (wid + 4 * j) * 8 + l8;
    avo[j] = (unsigned)(c * T + 4 * l7) * 4u | (c < C ? 0u : 0x80000000u);
    adst[j] = (unsigned)(c * G::APITCH + 8 * l7);
  }
  u32x4 ar[8];
  auto fetch_a = [&](int kb) __attribute__((always_inline)) {
    const unsigned soff = (unsigned)(b * C * T + kb * 32) * 4u;
    const unsigned dead = kb * 32 + 4 * l7 < T ? 0u : 0x80000000u;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      ar[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(avo[j] | dead), (int)soff, 0));
  };
  auto stage_a = [&](unsigned char* buf, float s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) put(buf, G::AP, adst[j], ar[j], s);
  };
  const int fk = lane >> 5, fc = lane & 31;
  floatx16 oacc[2][2];
  auto mma_b = [&](const unsigned char* abuf, int kb) __attribute__((always_inline)) {
    if (2 * wid >= cw) return;                                 // wave-uniform: no live channels in this wave's rows
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      halfx8 af[2][2], bf[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          af[cb][p] = *reinterpret_cast<const halfx8*>(abuf + p * G::AP + ((2 * wid + cb) * 32 + fc) * G::APITCH + (kk * 16 + 8 * fk) * 2);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          bf[nb][p] = *reinterpret_cast<const halfx8*>(bm + p * G::BP + (nb * 32 + fc) * G::BPITCH + (kb * 32 + kk * 16 + 8 * fk) * 2);
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            oacc[cb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb][SA[pr]], bf[nb][SB[pr]], oacc[cb][nb], 0, 0, 0);
    }
  };
  // probabilities (or score gradients) of this thread's accumulator tiles -> bm[n][m], two planes
  auto put_bm = [&](const floatx16 (&d)[MBW][2], float s) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v[4] = {s * d[mb][nb][4 * j], s * d[mb][nb][4 * j + 1], s * d[mb][nb][4 * j + 2], s * d[mb][nb][4 * j + 3]};
          u32x2 hi, lo;
          split4(v, hi, lo);
          const unsigned o = (unsigned)((nb * 32 + fc) * G::BPITCH + ((wid * MBW + mb) * 32 + 8 * j + 4 * fk) * 2);
          *reinterpret_cast<u32x2*>(bm + o) = hi;
          *reinterpret_cast<u32x2*>(bm + G::BP + o) = lo;
        }
  };
  // one GEMM of phase B: out[c, n] = beta out + post[n] / (sa sb) sum_m A[c, m] bm[n, m]
  auto phase_b = [&](int which, float s_a, float s_b, const float (&post)[2]) __attribute__((always_inline)) {
    ars = make_rsrc(a.a[which], tensor_bytes);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[cb][nb][e] = 0.f;
    fetch_a(0);
    stage_a(lds, s_a);
    if (KB > 1) fetch_a(1);
    __syncthreads();                                           // also: bm is complete
    for (int kb = 0; kb < KB; ++kb) {
      if (kb + 1 < KB) stage_a(lds + ((kb + 1) & 1) * G::ASTAGE, s_a);
      if (kb + 2 < KB) fetch_a(kb + 2);
      mma_b(lds + (kb & 1) * G::ASTAGE, kb);
      __syncthreads();
    }
    if (2 * wid >= cw) return;
    float* out = a.out[which] + (long)b * C * T;
    const float beta = a.beta[which];
    const float inv = 1.f / (s_a * s_b);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = n0 + nb * 32 + fc;
      if (n >= T) continue;
      const float f = inv * post[nb];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int cbase = (2 * wid + cb) * 32 + 4 * fk;
        if (cbase >= C) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float* p = out + (long)(cbase + acc_row(e)) * T + n;
          const float v = f * oacc[cb][nb][e];
          *p = beta != 0.f ? __fmaf_rn(beta, *p, v) : v;
        }
      }
    }
  };

  // column statistics over m: this thread's MBW x 16 values of column (nb, fc), its partner half-wave, the four waves
  auto col_reduce = [&](float (&v)[2], int slot, bool is_max) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float o = __shfl_xor(v[nb], 32, 64);
      v[nb] = is_max ? fmaxf(v[nb], o) : v[nb] + o;
    }
    if (fk == 0) { red[slot][wid][fc] = v[0]; red[slot][wid][32 + fc] = v[1]; }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float r0 = red[slot][0][nb * 32 + fc], r1 = red[slot][1][nb * 32 + fc];
      const float r2 = red[slot][2][nb * 32 + fc], r3 = red[slot][3][nb * 32 + fc];
      v[nb] = is_max ? fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)) : (r0 + r1) + (r2 + r3);
    }
  };
  const float ones[2] = {1.f, 1.f};
  const float P_SCALE = 8192.f;                                // probabilities are <= 1: 2^13

  if (MODE == MODE_FWD) {
    const float us = a.scale / (sx[0] * sy[0]);
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = (wid * MBW + mb) * 32 + 4 * fk + acc_row(e);
          const float s = m < T ? us * acc[0][mb][nb][e] : -INFINITY;
          acc[0][mb][nb][e] = s;
          mx[nb] = fmaxf(mx[nb], s);
        }
    col_reduce(mx, 0, true);
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = expf(acc[0][mb][nb][e] - mx[nb]);
          acc[0][mb][nb][e] = p;
          sum[nb] += p;
        }
    col_reduce(sum, 1, false);
    if (wid == 0 && fk == 0) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        if (n0 + nb * 32 + fc < T) a.lse[(long)b * T + n0 + nb * 32 + fc] = mx[nb] + logf(sum[nb]);
    }
    put_bm(acc[0], P_SCALE);
    const float post[2] = {1.f / sum[0], 1.f / sum[1]};
    phase_b(0, sa[0], P_SCALE, post);
  } else {
    // p = exp(scale s - lse), ds = scale p (dp - delta): lse / delta per column (DQ) or per row (DKV)
    const float us = a.scale / (sx[0] * sy[0]), ud = 1.f / (sx[1] * sy[1]);
    float lse_n[2] = {0.f, 0.f}, dl[2] = {0.f, 0.f};
    const bool nlive[2] = {n0 + fc < T, n0 + 32 + fc < T};     // columns beyond T (the last tile of a short sequence)
    if (MODE == MODE_DQ) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int n = n0 + nb * 32 + fc;
        lse_n[nb] = n < T ? a.lse[(long)b * T + n] : 0.f;
      }
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = (wid * MBW + mb) * 32 + 4 * fk + acc_row(e);
            const float p = (m < T && nlive[nb]) ? expf(us * acc[0][mb][nb][e] - lse_n[nb]) : 0.f;
            const float dp = ud * acc[1][mb][nb][e];
            acc[0][mb][nb][e] = p;
            acc[1][mb][nb][e] = dp;
            dl[nb] = __fmaf_rn(p, dp, dl[nb]);
          }
      col_reduce(dl, 0, false);
      if (wid == 0 && fk == 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          if (n0 + nb * 32 + fc < T) a.delta[(long)b * T + n0 + nb * 32 + fc] = dl[nb];
      }
    }
    float dmax = 0.f;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float p, dp, de;
          if (MODE == MODE_DQ) {
            p = acc[0][mb][nb][e]; dp = acc[1][mb][nb][e]; de = dl[nb];
          } else {
            const int m = (wid * MBW + mb) * 32 + 4 * fk + acc_row(e);
            p = nlive[nb] ? expf(us * acc[0][mb][nb][e] - rowstat[0][m]) : 0.f;
            dp = ud * acc[1][mb][nb][e]; de = rowstat[1][m];
            acc[0][mb][nb][e] = p;
          }
          const float ds = a.scale * (p * (dp - de));
          acc[1][mb][nb][e] = ds;
          dmax = fmaxf(dmax, fabsf(ds));
        }
    dmax = wave_max(dmax);
    if (lane == 0) sred[wid] = dmax;
    __syncthreads();
    const float sds = pow2_scale_of(fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3])));
    if (MODE == MODE_DQ) {
      put_bm(acc[1], sds);
      phase_b(0, sa[0], sds, ones);                            // dq = k ds
    } else {
      put_bm(acc[0], P_SCALE);
      phase_b(0, sa[0], P_SCALE, ones);                        // dv = do p      (ends with a barrier: bm is free again)
      put_bm(acc[1], sds);
      phase_b(1, sa[1], sds, ones);                            // dk = q ds
    }
  }
}

inline bool attn_ok(int B, int C, int T) {
  return B > 0 && C >= 32 && C <= 256 && C % 32 == 0 && T >= 4 && T <= 256 && T % 4 == 0 && (long)B * C * T * 4 < 0x7fffffffL;
}

template <int MODE>
int launch(const Args& a, hipStream_t s) {
  const unsigned grid = (unsigned)(a.B * ((a.T + 63) / 64));
  if (a.T <= 128) hipLaunchKernelGGL((attn_kernel<MODE, 128>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_kernel<MODE, 256>), dim3(grid), dim3(256), 0, s, a);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

}  // namespace

extern "C" {

int stk_attention_ok(int B, int C, int T) { return attn_ok(B, C, T) ? 1 : 0; }

int stk_attention_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse, float* rec, int B, int C,
                          int T, float scale, void* stream) {
  if (!q || !k || !v || !o || !lse || !rec || B <= 0 || C <= 0 || T <= 0) return STK_EINVAL;
  if (!attn_ok(B, C, T) || !stk_aligned16(q) || !stk_aligned16(k) || !stk_aligned16(v)) return STK_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  AmaxArgs m = {};
  m.x[0] = q; m.x[1] = k; m.x[2] = v; m.rec[0] = rec; m.rec[1] = rec + NPART; m.rec[2] = rec + 2 * NPART; m.n = (long)B * C * T;
  hipLaunchKernelGGL(attn_amax_kernel, dim3(NPART, 3), dim3(1024), 0, s, m);
  STK_CHECK_LAUNCH();
  Args a = {};
  a.x[0] = k; a.rx[0] = rec + NPART; a.y[0] = q; a.ry[0] = rec; a.a[0] = v; a.ra[0] = rec + 2 * NPART;
  a.x[1] = k; a.rx[1] = rec + NPART; a.y[1] = q; a.ry[1] = rec; a.a[1] = v; a.ra[1] = rec + 2 * NPART;
  a.out[0] = o; a.out[1] = o; a.beta[0] = a.beta[1] = 0.f; a.lse = lse; a.delta = nullptr;
  a.B = B; a.C = C; a.T = T; a.scale = scale;
  return launch<MODE_FWD>(a, s);
}

int stk_attention_bwd_f32(const float* q, const float* k, const float* v, const float* d_o, const float* lse, float* rec,
                          float* delta, float* dq, float beta_q, float* dk, float beta_k, float* dv, float beta_v, int B,
                          int C, int T, float scale, void* stream) {
  if (!q || !k || !v || !d_o || !lse || !rec || !delta || !dq || !dk || !dv || B <= 0 || C <= 0 || T <= 0) return STK_EINVAL;
  if (!attn_ok(B, C, T) || !stk_aligned16(q) || !stk_aligned16(k) || !stk_aligned16(v) || !stk_aligned16(d_o))
    return STK_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* rq = rec; float* rk = rec + NPART; float* rv = rec + 2 * NPART; float* rdo = rec + 3 * NPART;
  AmaxArgs m = {};
  m.x[0] = d_o; m.rec[0] = rdo; m.n = (long)B * C * T;
  hipLaunchKernelGGL(attn_amax_kernel, dim3(NPART, 1), dim3(1024), 0, s, m);
  STK_CHECK_LAUNCH();
  Args a = {};
  a.B = B; a.C = C; a.T = T; a.scale = scale; a.lse = const_cast<float*>(lse); a.delta = delta;
  // DQ: rows m = keys: X = (k, v); columns n = queries: Y = (q, do); phase B: dq = k ds
  a.x[0] = k; a.rx[0] = rk; a.x[1] = v; a.rx[1] = rv; a.y[0] = q; a.ry[0] = rq; a.y[1] = d_o; a.ry[1] = rdo;
  a.a[0] = k; a.ra[0] = rk; a.a[1] = k; a.ra[1] = rk; a.out[0] = dq; a.out[1] = dq; a.beta[0] = a.beta[1] = beta_q;
  int rc = launch<MODE_DQ>(a, s);
  if (rc) return rc;
  // DKV: rows m = queries: X = (q, do); columns n = keys: Y = (k, v); phase B: dv = do p, dk = q ds
  a.x[0] = q; a.rx[0] = rq; a.x[1] = d_o; a.rx[1] = rdo; a.y[0] = k; a.ry[0] = rk; a.y[1] = v; a.ry[1] = rv;
  a.a[0] = d_o; a.ra[0] = rdo; a.a[1] = q; a.ra[1] = rq; a.out[0] = dv; a.out[1] = dk; a.beta[0] = beta_v; a.beta[1] = beta_k;
  return launch<MODE_DKV>(a, s);
}

}  // extern "C"
