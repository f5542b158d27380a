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
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((__vector_size__(8)));
typedef __attribute__((address_space(3))) s4 lds_s4;

#ifndef ATTN_EXPERIMENT
#define ATTN_EXPERIMENT 0          // development: 1 = phase A without MFMAs, 2 = without staging, 3 = without fragment reads
#endif
constexpr int NPART = 256;          // partial |x| maxima per tensor (a "scale record", include/stk.h)
constexpr int MODE_FWD = 0, MODE_DQ = 1, MODE_DKV = 2;

// ---- |x| maxima of up to four tensors of n floats each in one launch: blockIdx.y picks the tensor ----------------------
struct AmaxArgs { const float* x[4]; float* rec[4]; long bs4[4]; long ct4; long n4; };    // tensor t: B segments of ct4 float4
__global__ __launch_bounds__(1024) void attn_amax_kernel(AmaxArgs a) {                     // bs4[t] float4 apart
  __shared__ float red[16];
  const float4* x4 = reinterpret_cast<const float4*>(a.x[blockIdx.y]);
  const long bs4 = a.bs4[blockIdx.y], ct4 = a.ct4;
  float m0 = 0.f, m1 = 0.f;
  const long stride = (long)NPART * 1024;
  auto at = [&](long i) { const long b = i / ct4; return x4[b * bs4 + (i - b * ct4)]; };
  long i = (long)blockIdx.x * 1024 + threadIdx.x;
  for (; i + stride < a.n4; i += 2 * stride) {
    const float4 u = at(i), v = at(i + stride);
    m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
    m1 = fmaxf(fmaxf(m1, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (i < a.n4) {
    const float4 u = at(i);
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
  long xs[2], ys[2], as[2], os[2];                               // floats between consecutive images of each operand
  float* lse;             // [B, T]  FWD: written; DQ / DKV: read
  float* delta;           // [B, T]  DQ: written; DKV: read
  int B, C, T; float scale;
  long long* dbg;         // development aid: cycle stamps of workgroup 0 / wave 0 at the phase boundaries (NULL: off)
};

template <int MODE, int TK>
struct Geo {
  static constexpr int NW = TK / 32;                      // waves: one 32-row block of D per wave (8 for TK = 256, 4 for 128)
  static constexpr int NT = 64 * NW;
  static constexpr int NG = MODE == MODE_FWD ? 1 : 2;     // GEMMs of phase A
  static constexpr int CK = 32 / NG;                      // channels per phase A chunk
  static constexpr int KS = CK / 16;                      // MFMA k steps per chunk
  static constexpr int XP = TK * CK * 2;                  // bytes per plane of an X chunk tile [m block][c][32]
  static constexpr int YP = 64 * CK * 2;
  static constexpr int GSZ = 2 * XP + 2 * YP;             // one GEMM's tiles: X hi, X lo, Y hi, Y lo
  static constexpr int STAGE = NG * GSZ;                  // 40960 (TK = 256) / 24576 (TK = 128)
  static constexpr int XI = CK / 8;                       // 16-byte pieces per thread, chunk and X operand (its wave's 32-block)
  static constexpr int YT = NG * 2 * (CK / 8) / NW;       // ... of the Y operands together (1 with 8 waves, 2 with 4)
  static constexpr int NPA = NG * XI + YT;                // pieces per thread and chunk
  static constexpr int CBW = 8 / NW;                      // phase B: 32-channel blocks of the output per wave
  static constexpr int AI = 32 / NW;                      // phase B: pieces per thread and 32-position block of A
  static constexpr int APITCH = 80;                       // phase B: A block [256 c][32 k] fp16, padded rows
  static constexpr int AP = 256 * APITCH;
  static constexpr int ASTAGE = 2 * AP;                   // 40960
  static constexpr int BPITCH = TK * 2 + 16;              // phase B: probabilities [64 n][TK k] fp16, padded rows
  static constexpr int BP = 64 * BPITCH;
  static constexpr int REGION0 = 2 * STAGE > 2 * ASTAGE ? 2 * STAGE : 2 * ASTAGE;     // 81920
  static constexpr int LDS = REGION0 + 2 * BP;
};

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// FULL: C == 256 and T == TK (every 16 x 16 attention block of the shipped configs).  Both loops are then unrolled into
// straight-line code -- not for the branches: with the chunk loop rolled, hipcc's s_waitcnt insertion merges the counter
// states of the loop's paths and makes every iteration wait for the loads it has just issued (vmcnt(9) ... vmcnt(0)
// behind a fresh 10-load fetch: 3000 cycles per chunk instead of 800) -- and the m < T masks disappear.
//
// Waves.  TK / 32 waves, two per SIMD for T = 256: a wave alone on its SIMD exposes every latency of its in-order
// stream (fragment reads, LDS store hand-off, barrier) and hides at most ~5 issue slots behind an MFMA; measured with
// four waves, a chunk took 2400 cycles of which the MFMAs are 768.  With a partner wave on the SIMD the two streams
// fill each other's gaps.
template <int MODE, int TK, bool FULL>
__global__ __launch_bounds__(TK * 2, TK / 128) void attn_kernel(Args a) {   // NT threads, NW / 4 waves per SIMD
  using G = Geo<MODE, TK>;
  constexpr int NG = G::NG, CK = G::CK, KS = G::KS, NW = G::NW, NT = G::NT;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS];
  __shared__ float red[2][NW][64];                // cross-wave column statistics
  __shared__ float rowstat[2][TK];                // DKV: lse[m], delta[m]
  __shared__ float sred[NW];
  unsigned char* const bm = lds + G::REGION0;     // probabilities / score gradients for phase B

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntile = FULL ? TK / 64 : (a.T + 63) >> 6;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int b = id / ntile, n0 = (id - b * ntile) * 64;
  const int T = FULL ? TK : a.T, C = FULL ? 256 : a.C;
  int stamp_i = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[stamp_i] = (long long)__builtin_amdgcn_s_memtime();
    ++stamp_i;
  };
  stamp();

  // ---- scales of the tensor operands: maxima of their 256-entry records -------------------------------------------------
  float sx[2] = {1.f, 1.f}, sy[2] = {1.f, 1.f}, sa[2] = {1.f, 1.f};
  {
    float m[6];
    const int ri = tid & 255;
    m[0] = a.rx[0][ri]; m[1] = a.ry[0][ri]; m[2] = a.ra[0][ri];
    m[3] = NG == 2 ? a.rx[1][ri] : 0.f; m[4] = NG == 2 ? a.ry[1][ri] : 0.f; m[5] = MODE == MODE_DKV ? a.ra[1][ri] : 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = wave_max(m[i]);
    float* r6 = reinterpret_cast<float*>(lds);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) r6[wid * 6 + i] = m[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) m[i] = fmaxf(fmaxf(r6[i], r6[6 + i]), fmaxf(r6[12 + i], r6[18 + i]));   // waves 0..3 hold all 256
    __syncthreads();
    sx[0] = pow2_scale_of(m[0]); sy[0] = pow2_scale_of(m[1]); sa[0] = pow2_scale_of(m[2]);
    sx[1] = pow2_scale_of(m[3]); sy[1] = pow2_scale_of(m[4]); sa[1] = pow2_scale_of(m[5]);
  }
  if (MODE == MODE_DKV) {
    for (int m = tid; m < TK; m += NT) {
      rowstat[0][m] = m < T ? a.lse[(long)b * T + m] : INFINITY;       // exp(s - inf) = 0 for the padding rows
      rowstat[1][m] = m < T ? a.delta[(long)b * T + m] : 0.f;
    }
  }

  stamp();
  // ---- phase A: D_g[m, n] = sum_c X_g[c, m] Y_g[c, n];  wave w owns rows m = 32 w .. 32 w + 31 -------------------------
  auto bytes_of = [&](long bstride) { return ((long)(a.B - 1) * bstride + (long)C * T) * 4; };
  const int l8 = lane >> 3, l7 = lane & 7;
  __amdgpu_buffer_rsrc_t xrs[NG], yrs[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) { xrs[g] = make_rsrc(a.x[g], bytes_of(a.xs[g])); yrs[g] = make_rsrc(a.y[g], bytes_of(a.ys[g])); }
  // A 16-byte load covers 4 positions of one channel; a wave instruction 8 channels x 32 positions (one 32-block).
  // X tiles: wave w stages 32-block w (the rows it multiplies), channel groups j = 0 .. CK / 8 - 1.
  // Y tiles (2 blocks x CK / 8 groups x NG operands = NW YT instructions): flat index f = wid + NW j.
  unsigned xvo[G::XI], xdst[G::XI], yvo[G::YT], ydst[G::YT];
  int yg[G::YT];
#pragma unroll
  for (int j = 0; j < G::XI; ++j) {
    const int c = j * 8 + l8, m = wid * 32 + 4 * l7;
    xvo[j] = (unsigned)(c * T + m) * 4u | (m < T ? 0u : 0x80000000u);          // bit 31: out of range -> zeros
    xdst[j] = (unsigned)((wid * CK + c) * 64 + 8 * l7);
  }
#pragma unroll
  for (int j = 0; j < G::YT; ++j) {
    const int f = wid + NW * j, per = 2 * (CK / 8), g = f / per, r = f % per;
    const int nblk = r & 1, c = (r >> 1) * 8 + l8, n = nblk * 32 + 4 * l7;
    yg[j] = g;                                                                  // wave-uniform
    yvo[j] = (unsigned)(c * T + n0 + n) * 4u | (n0 + n < T ? 0u : 0x80000000u);
    ydst[j] = (unsigned)(g * G::GSZ + 2 * G::XP + (nblk * CK + c) * 64 + 8 * l7);
  }
  // Two register sets: the loads of chunk c + 2 are issued at the top of iteration c while chunk c + 1 (the other set)
  // is being split and written to LDS between the MFMAs of chunk c.
  constexpr int NPA = G::NPA;
  u32x4 ra[2][NPA];
  auto fetch = [&](auto SET, int chunk) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const unsigned coff = (unsigned)(chunk * CK * T) * 4u;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int j = 0; j < G::XI; ++j)
        ra[S][g * G::XI + j] = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs[g], (int)xvo[j], (int)((unsigned)(b * a.xs[g]) * 4u + coff), 0));
#pragma unroll
    for (int j = 0; j < G::YT; ++j) {
      const bool second = NG == 2 && yg[j];
      ra[S][NG * G::XI + j] = __builtin_bit_cast(
          u32x4, __builtin_amdgcn_raw_buffer_load_b128(second ? yrs[NG - 1] : yrs[0], (int)yvo[j],
                                                       (int)((unsigned)(b * (second ? a.ys[NG - 1] : a.ys[0])) * 4u + coff), 0));
    }
  };
  // A piece goes to LDS in two halves (so that they can be placed between MFMAs): scale + hi terms, then lo terms + stores
  // (instruction selection is pinned by inline asm: left to itself the compiler packs the multiplies into v_pk_mul_f32
  // and evaluates the lo terms as v_cvt_f32_f16 + v_pk_add_f32 + s_nop, 12+ issue slots per piece instead of 10, and
  // packed fp32 VALU beside MFMAs costs extra -- MI355X_MICROARCH.md, price of a filler)
  struct Half { float v[4]; u32x2 hi; };
  auto half_a = [&](Half& h, u32x4 r, float s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_mul_f32 %0, %1, %2" : "=v"(h.v[i]) : "v"(s), "v"(__uint_as_float(r[i])));
    h.hi = u32x2{pack_h2(h.v[0], h.v[1]), pack_h2(h.v[2], h.v[3])};
  };
  auto half_b = [&](const Half& h, unsigned char* hi_plane, int plane_bytes, unsigned dst) __attribute__((always_inline)) {
    u32x2 lo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned d;                                  // lo = fp16(v - hi): one v_fma_mix per element, (-1) * hi + v
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h.hi[i]), "v"(h.v[2 * i]));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h.hi[i]), "v"(h.v[2 * i + 1]));
      lo[i] = d;
    }
    *reinterpret_cast<u32x2*>(hi_plane + dst) = h.hi;
    *reinterpret_cast<u32x2*>(hi_plane + plane_bytes + dst) = lo;
  };
  // piece q of a chunk: the X pieces of GEMM 0, of GEMM 1, then the Y pieces
  auto piece_a = [&](auto SET, int q, Half& h) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    float s;
    if (q < NG * G::XI) s = sx[q / G::XI];
    else s = (NG == 2 && yg[q - NG * G::XI]) ? sy[NG - 1] : sy[0];
    half_a(h, ra[S][q], s);
  };
  auto piece_b = [&](int q, const Half& h, unsigned char* buf) __attribute__((always_inline)) {
    if (q < NG * G::XI) half_b(h, buf + (q / G::XI) * G::GSZ, G::XP, xdst[q % G::XI]);
    else half_b(h, buf, G::YP, ydst[q - NG * G::XI]);
  };
  auto stage = [&](auto SET, unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NPA; ++q) { Half h; piece_a(SET, q, h); piece_b(q, h, buf); }
  };

  floatx16 acc[NG][2];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[g][nb][e] = 0.f;

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
  // One chunk: NMA MFMAs in (k step, GEMM) groups of 6, with the 2 NPA half pieces of the NEXT chunk's staging placed
  // between them (the scheduler left to itself runs all the staging first and the MFMAs back to back after it).
  constexpr int GS = 6, NMA = KS * NG * GS;
  constexpr int PRE_A = (2 * NPA) / 5, REST_A = 2 * NPA - PRE_A;
  auto chunk_a = [&](auto PAR, auto MORE, bool fetch2, int c) __attribute__((always_inline)) {
    constexpr int P = decltype(PAR)::value;
    constexpr bool more = decltype(MORE)::value;               // a next chunk exists: stage it between the MFMAs
    using SetNext = std::integral_constant<int, 1 - P>;
    const unsigned char* cur = lds + P * G::STAGE;
    unsigned char* nxt = lds + (1 - P) * G::STAGE;
    if (fetch2) fetch(PAR, c + 2);                            // set P was consumed when chunk c was staged
    halfx8 xa[KS][NG][2], yb[KS][NG][2][2];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          xa[kk][g][p] = tr8(cur + g * G::GSZ + p * G::XP + (wid * CK + kk * 16) * 64);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            yb[kk][g][nb][p] = tr8(cur + g * G::GSZ + 2 * G::XP + p * G::YP + (nb * CK + kk * 16) * 64);
        }
    Half h;
    int hp = 0;                                                 // half pieces done (compile-time after unrolling)
    auto half_piece = [&]() __attribute__((always_inline)) {
      if (more) {
        if ((hp & 1) == 0) piece_a(SetNext{}, hp >> 1, h); else piece_b(hp >> 1, h, nxt);
      }
      ++hp;
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PRE_A; ++i) half_piece();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NMA; ++i) {
      const int gi = i / GS, kk = gi / NG, g = gi % NG, r = i % GS, pr = r / 2, nb = r % 2;
      acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[kk][g][SA[pr]], yb[kk][g][nb][SB[pr]], acc[g][nb], 0, 0, 0);
      if (((i + 1) * REST_A) / NMA > (i * REST_A) / NMA) half_piece();
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nchunk = C / CK;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  fetch(P0{}, 0);
  if (nchunk > 1) fetch(P1{}, 1);
  stage(P0{}, lds);
  __syncthreads();
  if constexpr (FULL) {
    constexpr int NCH = 256 / CK;
    static_for<NCH>([&](auto ci) __attribute__((always_inline)) {
      constexpr int c = decltype(ci)::value;
      chunk_a(std::integral_constant<int, (c & 1)>{}, std::bool_constant<(c + 1 < NCH)>{}, c + 2 < NCH, c);
      __syncthreads();
    });
  } else {
    for (int c = 0; c < nchunk; c += 2) {
      if (c + 1 < nchunk) chunk_a(P0{}, std::true_type{}, c + 2 < nchunk, c); else chunk_a(P0{}, std::false_type{}, false, c);
      __syncthreads();
      if (c + 1 < nchunk) {
        if (c + 2 < nchunk) chunk_a(P1{}, std::true_type{}, c + 3 < nchunk, c + 1); else chunk_a(P1{}, std::false_type{}, false, c + 1);
        __syncthreads();
      }
    }
  }

  stamp();
  // ---- phase B plumbing: out[c, n] = sum_k A[c, k] bm[n, k];  wave w owns channel blocks CBW w .. CBW w + CBW - 1 ------
  constexpr int CBW = G::CBW, AI = G::AI;
  __amdgpu_buffer_rsrc_t ars = make_rsrc(a.a[0], bytes_of(a.as[0]));
  long a_bs = a.as[0];
  const int KB = FULL ? TK / 32 : (T + 31) >> 5;              // 32-position blocks of the contraction
  unsigned avo[AI], adst[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int c = (wid + NW * j) * 8 + l8;
    avo[j] = (unsigned)(c * T + 4 * l7) * 4u | (c < C ? 0u : 0x80000000u);
    adst[j] = (unsigned)(c * G::APITCH + 8 * l7);
  }
  u32x4 rb[2][AI];
  auto fetch_a = [&](auto SET, int kb) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const unsigned soff = (unsigned)(b * a_bs + kb * 32) * 4u;
    const unsigned dead = (FULL || kb * 32 + 4 * l7 < T) ? 0u : 0x80000000u;
#pragma unroll
    for (int j = 0; j < AI; ++j)
      rb[S][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(avo[j] | dead), (int)soff, 0));
  };
  const int fk = lane >> 5, fc = lane & 31;
  floatx16 oacc[CBW][2];
  constexpr int NMB = 12 * CBW, PRE_B = (2 * AI) / 5, REST_B = 2 * AI - PRE_B;
  // One 32-position block of phase B, the next block's pieces staged between the MFMAs
  auto block_b = [&](auto PAR, auto MORE, bool fetch2, int kb, float s_a) __attribute__((always_inline)) {
    constexpr int P = decltype(PAR)::value;
    constexpr bool more = decltype(MORE)::value;
    const unsigned char* abuf = lds + P * G::ASTAGE;
    unsigned char* nxt = lds + (1 - P) * G::ASTAGE;
    if (fetch2) fetch_a(PAR, kb + 2);
    halfx8 af[2][CBW][2], bf[2][2][2];                          // [k step][block][plane]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
          af[kk][cb][p] = *reinterpret_cast<const halfx8*>(abuf + p * G::AP + ((CBW * wid + cb) * 32 + fc) * G::APITCH + (kk * 16 + 8 * fk) * 2);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          bf[kk][nb][p] = *reinterpret_cast<const halfx8*>(bm + p * G::BP + (nb * 32 + fc) * G::BPITCH + (kb * 32 + kk * 16 + 8 * fk) * 2);
      }
    Half h;
    int hp = 0;
    auto half_piece = [&]() __attribute__((always_inline)) {
      if (more) {
        if ((hp & 1) == 0) half_a(h, rb[1 - P][hp >> 1], s_a); else half_b(h, nxt, G::AP, adst[hp >> 1]);
      }
      ++hp;
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PRE_B; ++i) half_piece();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NMB; ++i) {
      const int kk = i / (6 * CBW), r = i % (6 * CBW), pr = r / (2 * CBW), cb = (r / 2) % CBW, nb = r % 2;
      oacc[cb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][cb][SA[pr]], bf[kk][nb][SB[pr]], oacc[cb][nb], 0, 0, 0);
      if (((i + 1) * REST_B) / NMB > (i * REST_B) / NMB) half_piece();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // probabilities (or score gradients) of this thread's accumulator tiles -> bm[n][m], two planes
  auto put_bm = [&](const floatx16 (&d)[2], float s) __attribute__((always_inline)) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v[4] = {s * d[nb][4 * j], s * d[nb][4 * j + 1], s * d[nb][4 * j + 2], s * d[nb][4 * j + 3]};
        u32x2 hi, lo;
        split4(v, hi, lo);
        const unsigned o = (unsigned)((nb * 32 + fc) * G::BPITCH + (wid * 32 + 8 * j + 4 * fk) * 2);
        *reinterpret_cast<u32x2*>(bm + o) = hi;
        *reinterpret_cast<u32x2*>(bm + G::BP + o) = lo;
      }
  };
  // one GEMM of phase B: out[c, n] = beta out + post[n] / (sa sb) sum_m A[c, m] bm[n, m]
  auto phase_b = [&](int which, float s_a, float s_b, const float (&post)[2]) __attribute__((always_inline)) {
    ars = make_rsrc(a.a[which], bytes_of(a.as[which]));
    a_bs = a.as[which];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[cb][nb][e] = 0.f;
    fetch_a(P0{}, 0);
    if (KB > 1) fetch_a(P1{}, 1);
#pragma unroll
    for (int j = 0; j < AI; ++j) { Half h; half_a(h, rb[0][j], s_a); half_b(h, lds, G::AP, adst[j]); }
    __syncthreads();                                           // also: bm is complete
    stamp();
    if constexpr (FULL) {
      constexpr int NKB = TK / 32;
      static_for<NKB>([&](auto ki) __attribute__((always_inline)) {
        constexpr int kb = decltype(ki)::value;
        block_b(std::integral_constant<int, (kb & 1)>{}, std::bool_constant<(kb + 1 < NKB)>{}, kb + 2 < NKB, kb, s_a);
        __syncthreads();
      });
    } else {
      for (int kb = 0; kb < KB; kb += 2) {
        if (kb + 1 < KB) block_b(P0{}, std::true_type{}, kb + 2 < KB, kb, s_a); else block_b(P0{}, std::false_type{}, false, kb, s_a);
        __syncthreads();
        if (kb + 1 < KB) {
          if (kb + 2 < KB) block_b(P1{}, std::true_type{}, kb + 3 < KB, kb + 1, s_a); else block_b(P1{}, std::false_type{}, false, kb + 1, s_a);
          __syncthreads();
        }
      }
    }
    stamp();
    // rows of channels >= C were computed from zeros and are not stored
    __amdgpu_buffer_rsrc_t ors = make_rsrc(a.out[which], bytes_of(a.os[which]));
    const float beta = a.beta[which];
    const float inv = 1.f / (s_a * s_b);
    const unsigned obase = (unsigned)(b * a.os[which]) * 4u;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = n0 + nb * 32 + fc;
      const float f = inv * post[nb];
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) {
        const int cbase = (CBW * wid + cb) * 32 + 4 * fk;
        const unsigned vo = (unsigned)(cbase * T + n) * 4u | ((n < T && cbase < C) ? 0u : 0x80000000u);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned so = obase + (unsigned)(acc_row(e) * T) * 4u;
          float v = f * oacc[cb][nb][e];
          if (beta != 0.f)
            v = __fmaf_rn(beta, __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, (int)vo, (int)so, 0)), v);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, (int)vo, (int)so, 0);
        }
      }
    }
  };

  // column statistics over m: this thread's 16 values of column (nb, fc), its partner half-wave, the NW waves
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
      float r = red[slot][0][nb * 32 + fc];
#pragma unroll
      for (int w = 1; w < NW; ++w) r = is_max ? fmaxf(r, red[slot][w][nb * 32 + fc]) : r + red[slot][w][nb * 32 + fc];
      v[nb] = r;
    }
  };
  const float ones[2] = {1.f, 1.f};
  const float P_SCALE = 8192.f;                                // probabilities are <= 1: 2^13

  if (MODE == MODE_FWD) {
    const float us = a.scale / (sx[0] * sy[0]);
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = wid * 32 + 4 * fk + acc_row(e);
        const float s = (FULL || m < T) ? us * acc[0][nb][e] : -INFINITY;
        acc[0][nb][e] = s;
        mx[nb] = fmaxf(mx[nb], s);
      }
    col_reduce(mx, 0, true);
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = expf(acc[0][nb][e] - mx[nb]);
        acc[0][nb][e] = p;
        sum[nb] += p;
      }
    col_reduce(sum, 1, false);
    if (wid == 0 && fk == 0) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        if (n0 + nb * 32 + fc < T) a.lse[(long)b * T + n0 + nb * 32 + fc] = mx[nb] + logf(sum[nb]);
    }
    stamp();
    put_bm(acc[0], P_SCALE);
    const float post[2] = {1.f / sum[0], 1.f / sum[1]};
    phase_b(0, sa[0], P_SCALE, post);
    stamp();
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
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = wid * 32 + 4 * fk + acc_row(e);
          const float p = (FULL || (m < T && nlive[nb])) ? expf(us * acc[0][nb][e] - lse_n[nb]) : 0.f;
          const float dp = ud * acc[1][nb][e];
          acc[0][nb][e] = p;
          acc[1][nb][e] = dp;
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
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p, dp, de;
        if (MODE == MODE_DQ) {
          p = acc[0][nb][e]; dp = acc[1][nb][e]; de = dl[nb];
        } else {
          const int m = wid * 32 + 4 * fk + acc_row(e);
          p = (FULL || nlive[nb]) ? expf(us * acc[0][nb][e] - rowstat[0][m]) : 0.f;
          dp = ud * acc[1][nb][e]; de = rowstat[1][m];
          acc[0][nb][e] = p;
        }
        const float ds = a.scale * (p * (dp - de));
        acc[1][nb][e] = ds;
        dmax = fmaxf(dmax, fabsf(ds));
      }
    dmax = wave_max(dmax);
    if (lane == 0) sred[wid] = dmax;
    __syncthreads();
    float dm = sred[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) dm = fmaxf(dm, sred[w]);
    const float sds = pow2_scale_of(dm);
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
inline bool stride_ok(int B, int C, int T, long bs) {
  return bs >= (long)C * T && bs % 4 == 0 && ((long)(B - 1) * bs + (long)C * T) * 4 < 0x7fffffffL;
}

template <int MODE>
int launch(const Args& a, hipStream_t s) {
  const unsigned grid = (unsigned)(a.B * ((a.T + 63) / 64));
  if (a.T == 256 && a.C == 256) hipLaunchKernelGGL((attn_kernel<MODE, 256, true>), dim3(grid), dim3(512), 0, s, a);
  else if (a.T <= 128) hipLaunchKernelGGL((attn_kernel<MODE, 128, false>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_kernel<MODE, 256, false>), dim3(grid), dim3(512), 0, s, a);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

void fwd_args(Args& a, const float* q, const float* k, const float* v, long bs, float* o, float* lse, float* rec, int B, int C,
              int T, float scale) {
  const long ct = (long)C * T;
  a.x[0] = k; a.rx[0] = rec + NPART; a.y[0] = q; a.ry[0] = rec; a.a[0] = v; a.ra[0] = rec + 2 * NPART;
  a.x[1] = k; a.rx[1] = rec + NPART; a.y[1] = q; a.ry[1] = rec; a.a[1] = v; a.ra[1] = rec + 2 * NPART;
  a.xs[0] = a.xs[1] = a.ys[0] = a.ys[1] = a.as[0] = a.as[1] = bs;
  a.out[0] = o; a.out[1] = o; a.os[0] = a.os[1] = ct; a.beta[0] = a.beta[1] = 0.f; a.lse = lse; a.delta = nullptr;
  a.B = B; a.C = C; a.T = T; a.scale = scale;
}

}  // namespace

extern "C" {

int stk_attention_ok(int B, int C, int T) { return attn_ok(B, C, T) ? 1 : 0; }

int stk_attention_fwd_f32(const float* q, const float* k, const float* v, long qkv_bstride, float* o, float* lse, float* rec,
                          int B, int C, int T, float scale, void* stream) {
  if (!q || !k || !v || !o || !lse || !rec || B <= 0 || C <= 0 || T <= 0) return STK_EINVAL;
  if (!attn_ok(B, C, T) || !stride_ok(B, C, T, qkv_bstride) || !stk_aligned16(q) || !stk_aligned16(k) || !stk_aligned16(v))
    return STK_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  AmaxArgs m = {};
  m.x[0] = q; m.x[1] = k; m.x[2] = v; m.rec[0] = rec; m.rec[1] = rec + NPART; m.rec[2] = rec + 2 * NPART;
  m.bs4[0] = m.bs4[1] = m.bs4[2] = qkv_bstride / 4; m.ct4 = (long)C * T / 4; m.n4 = (long)B * m.ct4;
  hipLaunchKernelGGL(attn_amax_kernel, dim3(NPART, 3), dim3(1024), 0, s, m);
  STK_CHECK_LAUNCH();
  Args a = {};
  fwd_args(a, q, k, v, qkv_bstride, o, lse, rec, B, C, T, scale);
  return launch<MODE_FWD>(a, s);
}

/* development aid (not in include/stk.h): the forward kernel alone with cycle stamps of workgroup 0 in dbg[0..8) */
int stk_attention_fwd_debug(const float* q, const float* k, const float* v, float* o, float* lse, float* rec, int B, int C,
                            int T, float scale, long long* dbg, void* stream) {
  Args a = {};
  fwd_args(a, q, k, v, (long)C * T, o, lse, rec, B, C, T, scale);
  a.dbg = dbg;
  return launch<MODE_FWD>(a, (hipStream_t)stream);
}

int stk_attention_bwd_f32(const float* q, const float* k, const float* v, long qkv_bstride, const float* d_o, const float* lse,
                          float* rec, float* delta, float* dq, float beta_q, float* dk, float beta_k, float* dv, float beta_v,
                          long grad_bstride, int B, int C, int T, float scale, void* stream) {
  if (!q || !k || !v || !d_o || !lse || !rec || !delta || !dq || !dk || !dv || B <= 0 || C <= 0 || T <= 0) return STK_EINVAL;
  if (!attn_ok(B, C, T) || !stride_ok(B, C, T, qkv_bstride) || !stride_ok(B, C, T, grad_bstride) || !stk_aligned16(q) ||
      !stk_aligned16(k) || !stk_aligned16(v) || !stk_aligned16(d_o))
    return STK_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const long ct = (long)C * T;
  float* rq = rec; float* rk = rec + NPART; float* rv = rec + 2 * NPART; float* rdo = rec + 3 * NPART;
  AmaxArgs m = {};
  m.x[0] = d_o; m.rec[0] = rdo; m.bs4[0] = ct / 4; m.ct4 = ct / 4; m.n4 = (long)B * m.ct4;
  hipLaunchKernelGGL(attn_amax_kernel, dim3(NPART, 1), dim3(1024), 0, s, m);
  STK_CHECK_LAUNCH();
  Args a = {};
  a.B = B; a.C = C; a.T = T; a.scale = scale; a.lse = const_cast<float*>(lse); a.delta = delta;
  const long bs = qkv_bstride, gs = grad_bstride;
  // DQ: rows m = keys: X = (k, v); columns n = queries: Y = (q, do); phase B: dq = k ds
  a.x[0] = k; a.rx[0] = rk; a.xs[0] = bs; a.x[1] = v; a.rx[1] = rv; a.xs[1] = bs;
  a.y[0] = q; a.ry[0] = rq; a.ys[0] = bs; a.y[1] = d_o; a.ry[1] = rdo; a.ys[1] = ct;
  a.a[0] = k; a.ra[0] = rk; a.as[0] = bs; a.a[1] = k; a.ra[1] = rk; a.as[1] = bs;
  a.out[0] = dq; a.out[1] = dq; a.os[0] = a.os[1] = gs; a.beta[0] = a.beta[1] = beta_q;
  int rc = launch<MODE_DQ>(a, s);
  if (rc) return rc;
  // DKV: rows m = queries: X = (q, do); columns n = keys: Y = (k, v); phase B: dv = do p, dk = q ds
  a.x[0] = q; a.rx[0] = rq; a.xs[0] = bs; a.x[1] = d_o; a.rx[1] = rdo; a.xs[1] = ct;
  a.y[0] = k; a.ry[0] = rk; a.ys[0] = bs; a.y[1] = v; a.ry[1] = rv; a.ys[1] = bs;
  a.a[0] = d_o; a.ra[0] = rdo; a.as[0] = ct; a.a[1] = q; a.ra[1] = rq; a.as[1] = bs;
  a.out[0] = dv; a.out[1] = dk; a.os[0] = a.os[1] = gs; a.beta[0] = beta_v; a.beta[1] = beta_k;
  return launch<MODE_DKV>(a, s);
}

}  // extern "C"
