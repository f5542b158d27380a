// conv_x3.h -- 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) on the bf16 matrix pipe with a
// three-way operand split ("x3"), for gfx950.  Included by conv.hip inside its anonymous namespace.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate (157 TFLOP/s); the bf16 MFMA is 16x faster.  An fp32
// value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand bits, taken by truncation):
//     a = a0 + a1 + a2,   b = b0 + b1 + b2
//     a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + O(2^-24 |ab|)
// so six bf16 MFMAs (products exact, fp32 accumulate) give the fp32 product to fp32 rounding accuracy -- measured
// against the double-precision oracle the error is the same as (slightly below) the f32-input MFMA's, see
// tests/test_gpu_kernels.py -- at 6/16 of the matrix-pipe time.
//
// GEMM view (both directions share the kernel):   out[m, n] = sum_k  Wp[m, k] * act[k, n]
//     forward  m = co, n = (b, y, x), k = (tap, ci):   act = x[b, ci, y + kh - 1, x + kw - 1]
//     dgrad    m = ci, n = (b, y, x), k = (tap', co):  act = dy[b, co, y + kh' - 1, x + kw' - 1], tap' = 8 - tap
// K is ordered TAP-MAJOR so that a 32-wide k chunk is one tap and 32 consecutive channels: the tap (hence the
// halo test and the pixel shift) is uniform over the chunk and a thread's 16 loads differ only by a scalar
// channel-plane offset.
//
//   * weights are re-laid out and split ONCE per call by wprep_kernel into Wp[split][tap][row (padded to 128)][k]
//     bf16 (<= 14 MB, L2 resident): the A loader is six 16-byte copies per thread and chunk, no conversion;
//   * activations are split in the B loader when they are written to LDS (and/sub/perm, ~6 VALU per element);
//   * LDS tiles are [split][row][32 k] bf16 with an 80-byte row pitch: the MFMA operand reads (one ds_read_b128
//     per lane = 8 consecutive k of one row) and the staging writes are bank-conflict free;
//   * 128 x 128 tile, 4 waves of 64 x 64 (2 x 2 MFMA tiles): per 16-k step 12 ds_read_b128 and 24 MFMAs per wave,
//     small terms first so the accumulation order is fixed.
#pragma once

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace x3 {

constexpr int KC = 32;
constexpr int PITCH = 80;              // bytes per LDS row: 32 bf16 + 16 bytes of padding
constexpr int PLANE = 128 * PITCH;     // one split plane of one operand
constexpr int OPER = 3 * PLANE;
constexpr int LDS_BYTES = 2 * OPER;    // 61440

struct Src {             // the activation operand: channel-concat of two NCHW tensors
  const float* s1; const float* s2; int S1, S2;
  const unsigned short* wp; int Mpad; int Kc;     // prepared weights, padded row count, channels (= S1 + S2)
};

__device__ __forceinline__ float hi_part(float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); }
// pack the bf16 (upper) halves of two floats: low half <- lo, high half <- hi
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}

// Wp[split][tap][row][k] = split_s( transpose ? w[(k*Cin + row)*9 + 8 - tap] : w[(row*Cin + k)*9 + tap] ),
// rows >= the real row count are zero.  One thread per (row, k): reads its 9 taps (36 contiguous bytes).
__global__ __launch_bounds__(256) void wprep_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                    int Cout, int Cin, int Mpad, int transpose) {
  const int Kd = transpose ? Cout : Cin, R = transpose ? Cin : Cout;
  const long total = (long)Mpad * Kd;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int row = (int)(i / Kd), k = (int)(i - (long)row * Kd);
  float v[9];
  if (row < R) {
    const float* s = w + (transpose ? ((long)k * Cin + row) : ((long)row * Cin + k)) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = s[t];
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = 0.f;
  }
  const long plane = 9L * Mpad * Kd;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float a = v[transpose ? 8 - t : t];
    const float h0 = hi_part(a), r1 = a - h0;
    const float h1 = hi_part(r1), h2 = r1 - h1;           // h2 has <= 8 significant bits: exact in bf16
    const long o = ((long)t * Mpad + row) * Kd + k;
    out[o] = (unsigned short)(__float_as_uint(h0) >> 16);
    out[plane + o] = (unsigned short)(__float_as_uint(h1) >> 16);
    out[2 * plane + o] = (unsigned short)(__float_as_uint(h2) >> 16);
  }
}

template <class EP, bool DUAL>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvP p, Src q, int M, int Nn, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  unsigned char* As = lds;
  unsigned char* Bs = lds + OPER;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * 128;
  const int HW = p.HW, W = p.W;

  // ---- B loader state: this thread's pixel and its 9-tap halo mask
  const int nl = tid & 127;
  const int kg = __builtin_amdgcn_readfirstlane(tid >> 7);      // which 16 of the chunk's 32 channels
  unsigned mask = 0; int tb1 = 0, tb2 = 0;
  {
    const int n = n0 + nl;
    if (n < Nn) {
      const int b = n / HW, hw = n - b * HW;
      const int y = hw / W, x = hw - y * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < W) mask |= 1u << t;
      }
      tb1 = b * q.S1 * HW + hw;
      tb2 = b * q.S2 * HW + hw;
    }
  }
  // ---- A loader state: six 16-byte pieces per chunk: (split j>>1, row (tid>>2) + 64 (j&1), segment tid&3)
  const int a_row = tid >> 2, a_seg = tid & 3;
  const long a_plane = 9L * q.Mpad * q.Kc;                       // elements per split plane
  const unsigned short* a_base = q.wp + ((long)(m0 + a_row) * q.Kc + a_seg * 8);

  const int cpt = q.Kc / KC;          // chunks per tap
  const int nchunks = 9 * cpt;

  float rb[16]; u32x4 ra[6]; unsigned bok = 0;
  auto load = [&](int c) {
    const int tap = c / cpt, cc = c - tap * cpt;                 // scalar
    // weights
    const unsigned short* s = a_base + ((long)tap * q.Mpad * q.Kc + cc * KC);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      ra[j] = *reinterpret_cast<const u32x4*>(s + (j >> 1) * a_plane + (long)(j & 1) * 64 * q.Kc);
    // activations: 16 channels of one tap-shifted pixel; unconditional loads from a safe offset
    const int ci0 = cc * KC + kg * 16;
    const bool first = !DUAL || ci0 < q.S1;
    const uintptr_t tensor = first ? (uintptr_t)q.s1 : (uintptr_t)q.s2;
    const gfloat* plane = (const gfloat*)(tensor + (uintptr_t)(first ? ci0 : ci0 - q.S1) * (uintptr_t)HW * 4u);
    bok = (mask >> tap) & 1u;
    const int off = bok ? (first ? tb1 : tb2) + (tap / 3 - 1) * W + (tap % 3 - 1) : 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) rb[i] = (plane + (long)i * HW)[off];
  };
  auto store = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j)
      *reinterpret_cast<u32x4*>(As + (j >> 1) * PLANE + (a_row + 64 * (j & 1)) * PITCH + a_seg * 16) = ra[j];
    const unsigned zm = bok ? 0xffffffffu : 0u;
    unsigned pk[3][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float v0 = __uint_as_float(__float_as_uint(rb[2 * u]) & zm);
      const float v1 = __uint_as_float(__float_as_uint(rb[2 * u + 1]) & zm);
      const float r0 = v0 - hi_part(v0), r1 = v1 - hi_part(v1);
      const float t0 = r0 - hi_part(r0), t1 = r1 - hi_part(r1);
      pk[0][u] = pack_hi(v0, v1);
      pk[1][u] = pack_hi(r0, r1);
      pk[2][u] = pack_hi(t0, t1);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      unsigned char* d = Bs + s * PLANE + nl * PITCH + kg * 32;
      *reinterpret_cast<u32x4*>(d) = u32x4{pk[s][0], pk[s][1], pk[s][2], pk[s][3]};
      *reinterpret_cast<u32x4*>(d + 16) = u32x4{pk[s][4], pk[s][5], pk[s][6], pk[s][7]};
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm0 = (wid & 1) * 64, wn0 = (wid >> 1) * 64;
  const int fk = lane >> 5, fc = lane & 31;
  const unsigned char* a_rd = As + (wm0 + fc) * PITCH + fk * 16;
  const unsigned char* b_rd = Bs + (wn0 + fc) * PITCH + fk * 16;

  // Pipeline (LDS single-buffered, two barriers per chunk):
  //   B1: chunk c is in LDS          -> read the kk = 0 fragments, 24 MFMAs, read the kk = 1 fragments
  //   B2: nobody reads LDS any more  -> 24 MFMAs of kk = 1 INTERLEAVED with the split + LDS write of chunk c + 1
  //                                     (whose global loads were issued one full iteration earlier), then the
  //                                     global loads of chunk c + 2 are issued.
  // So the conversion VALU and the LDS stores run in the shadow of the matrix pipe of the same wave, and a global
  // load has a whole chunk period (~3k cycles) to land.
#define STK_X3_FRAGS(KK)                                                                               \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int s = 0; s < 3; ++s) {         \
    a[i][s] = *reinterpret_cast<const bf16x8*>(a_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);          \
    b[i][s] = *reinterpret_cast<const bf16x8*>(b_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);          \
  }
  // six products per tile, smallest terms first; the four tiles interleave so that consecutive MFMAs never
  // wait on the same accumulator
#define STK_X3_PROD(SA, SB)                                                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)            \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][SA], b[j][SB], acc[i][j], 0, 0, 0);
#define STK_X3_MFMAS STK_X3_PROD(2, 0) STK_X3_PROD(1, 1) STK_X3_PROD(0, 2) STK_X3_PROD(1, 0) STK_X3_PROD(0, 1) STK_X3_PROD(0, 0)
  load(0);
  store();
  load(min(1, nchunks - 1));
  bf16x8 a[2][3], b[2][3];
  for (int c = 0; c + 1 < nchunks; ++c) {
    __syncthreads();                                   // B1
    STK_X3_FRAGS(0)
    STK_X3_MFMAS
    STK_X3_FRAGS(1)
    __syncthreads();                                   // B2
    STK_X3_MFMAS
    store();                                           // chunk c + 1
    load(min(c + 2, nchunks - 1));                     // (the last iteration re-loads the last chunk: harmless)
    // one MFMA, then a slice of the staging work: ~150 VALU, 12 LDS writes, 22 global loads over 24 MFMAs
#pragma unroll
    for (int g = 0; g < 24; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);     // VALU
      if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
    }
  }
  __syncthreads();
  STK_X3_FRAGS(0)
  STK_X3_MFMAS
  STK_X3_FRAGS(1)
  STK_X3_MFMAS
#undef STK_X3_MFMAS
#undef STK_X3_PROD
#undef STK_X3_FRAGS

  EP ep;
  ep.init(p, 0, 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn0 + j * 32 + fc;
    const bool nok = n < Nn;
    ep.col(p, nok ? n : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) ep.strip(p, m0 + wm0 + i * 32 + 4 * fk, M, nok, nok ? n : 0, acc[i][j]);
  }
}

inline int pad128(int v) { return (v + 127) / 128 * 128; }
// bytes of prepared weights for an M x Kc 3x3 layer
inline long wp_bytes(int M, int Kc) { return 3L * 9 * pad128(M) * Kc * 2; }

}  // namespace x3
