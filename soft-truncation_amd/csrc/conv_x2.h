// conv_x2.h -- forward / data-gradient convolution (3x3 pad 1 and 1x1, stride 1) on the fp16 matrix pipe with a TWO-way
// operand split ("x2"), for gfx950.  Included by conv.hip inside its anonymous namespace, after conv_x3.h, whose tile
// geometry, buffer-load helpers and K order it shares.
//
// Why: the split kernels are limited by the matrix pipe's power-limited clock, i.e. by the number of MFMAs.  An fp32
// value is, to 2^-24 relative, the sum of two fp16 values (11 + 11 significand bits, round to nearest), so
//     a*b = a0*b0 + (a0*b1 + a1*b0) + O(2^-22 |ab|)
// needs THREE fp16 MFMAs where the bf16 three-way split of conv_x3.h needs six (8 + 8 + 8 bits).  fp16 has only 5
// exponent bits, so each operand tensor is first multiplied by the power of two that puts its largest magnitude in
// [2^13, 2^14) -- exact, and undone exactly in the epilogue.  An element more than 2^-16 below its tensor's maximum
// loses (part of) its second term to fp16's subnormal range; its error then stays below 2^-39 of that maximum.
// Measured against float64 (tools/split_accuracy.py, bench.py `arithmetic_check`): 0.8-1.6e-6 of max|y|, the same as the
// f32-input MFMA and the bf16 three-way split, also for gradient-like tensors whose images spread over four decades.
//
// Scales: the weights' |w| maximum is taken when they are prepared (wamax -> 256-byte header of the prepared block); an
// activation tensor's |x| maximum comes from amax_partial_kernel, launched by the conv call itself: 256 partial maxima
// that every workgroup of the consumer reduces on its own, so there is no finishing launch and no host round trip.
#pragma once

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));

namespace x2 {

using x3::KC;
using x3::PITCH;
using x3::PLANE;
constexpr int OPER = 2 * PLANE;
constexpr int LDS_BYTES = 2 * OPER;    // 40960
constexpr int NPART = 256;             // partial maxima per activation tensor
constexpr int HEADER = 256;            // bytes in front of the planes of a prepared block: WPART partial |w| maxima

// part[b] = max |x| over block b's grid-stride share (any n; 16-byte path when aligned).  1024 threads per block and two
// independent 16-byte loads per thread and trip: with 256 x 256 threads the pass ran at 2.6 TB/s, latency-bound.
constexpr int AMAX_THREADS = 1024;
__global__ __launch_bounds__(AMAX_THREADS) void amax_partial_kernel(const float* __restrict__ x, long n,
                                                                   float* __restrict__ part) {
  __shared__ float red[AMAX_THREADS / 64];
  float m0 = 0.f, m1 = 0.f;
  const long n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n >> 2 : 0;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const long stride = (long)NPART * AMAX_THREADS;
  long i = (long)blockIdx.x * AMAX_THREADS + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 u = x4[i], v = x4[i + stride];
    m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
    m1 = fmaxf(fmaxf(m1, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (i < n4) {
    const float4 u = x4[i];
    m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w)));
  }
  for (long j = 4 * n4 + (long)blockIdx.x * AMAX_THREADS + threadIdx.x; j < n; j += stride) m1 = fmaxf(m1, fabsf(x[j]));
  float m = wave_max(fmaxf(m0, m1));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < AMAX_THREADS / 64; ++w) m = fmaxf(m, red[w]);
    part[blockIdx.x] = m;
  }
}

// power of two s with m * s in [2^13, 2^14) (1 for m = 0)
__device__ __forceinline__ float pow2_scale_of(float m) {
  const int be = (int)((__float_as_uint(m) >> 23) & 0xffu);          // m = 1.f * 2^(be - 127)
  if (be == 0) return 1.f;
  const int se = min(max(127 + 13 - (be - 127), 1), 254);
  return __uint_as_float((unsigned)se << 23);
}
// block-wide maximum of `count` (256 or 512) partials; all 256 threads call it; `red` = 4 floats of LDS
__device__ __forceinline__ float block_amax(const float* __restrict__ part, int count, float* red) {
  float m = part[threadIdx.x];
  if (count > NPART) m = fmaxf(m, part[NPART + threadIdx.x]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return m;
}

// The same maximum with the LOADS at the top of a kernel and the reduction after its main loop (round 6): the scale of a plane operand
// is needed by the epilogue only, so its load latency (and the barriers of the reduction) need not stand in front of the first tile load --
// 2-3 us per workgroup round, which a 27 us K-split launch or a one-round weight gradient notices.  `red`: 4 floats of LDS of its own.
struct LateAmax {
  float m;
  __device__ __forceinline__ void load(const float* __restrict__ part, int count, int tid) {
    m = part[tid];
    if (count > NPART) m = fmaxf(m, part[NPART + tid]);
  }
  __device__ __forceinline__ float reduce(float* red, int tid) {            // all 256 threads of the (group of the) workgroup
    const float v = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
};

__device__ __forceinline__ unsigned pack_h2(float lo, float hi) {
  const halfx2 v = {(_Float16)lo, (_Float16)hi};                      // round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_part(float v) { return v - (float)(_Float16)v; }   // exact

// ---- weight preparation ---------------------------------------------------------------------------------------------
// A prepared block = [HEADER bytes: partial |w| maxima][2 planes: Wp[split][k / 32][tap][row (pad 128)][k % 32] fp16 of scale*W],
// W(row, k, tap) indexed as in x3::wprep_kernel.  WprepDesc == StkWprepDesc (include/stk.h); wp points at the header.
struct WprepDesc {
  const float* w; unsigned char* wp; long sm, sk; int M, Kc, Mpad, taps, flip, reserved;
};
// |w| maxima: WPART workgroups per layer (blockIdx.y = layer, blockIdx.x = share) write one partial each into the header;
// the consumers (wprep_kernel, gemm_kernel) take the maximum of the WPART floats.  No atomics, nothing to zero.
constexpr int WPART = 16;
__global__ __launch_bounds__(256) void wamax_kernel(const WprepDesc* __restrict__ descs, WprepDesc one) {
  __shared__ float red[4];
  const WprepDesc d = descs ? descs[blockIdx.y] : one;
  const long n = (long)d.M * d.Kc * d.taps;
  float m = 0.f;
  if ((reinterpret_cast<uintptr_t>(d.w) & 15) == 0) {
    // 16-byte loads, four in flight per thread (round 4: one 4-byte load per trip made the 16 workgroups of a 256 x 256 x 9
    // layer walk 144 dependent round trips each -- 186 us per step for 232 MB)
    const float4* w4 = reinterpret_cast<const float4*>(d.w);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long)WPART * 1024) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long j = i + 256 * u;
        v[u] = j < n4 ? w4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(d.w[(n4 << 2) + threadIdx.x]));
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)WPART * 256) m = fmaxf(m, fabsf(d.w[i]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) reinterpret_cast<float*>(d.wp)[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float weight_scale(const void* header) {
  const float* h = static_cast<const float*>(header);
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < WPART; ++i) m = fmaxf(m, h[i]);
  return pow2_scale_of(m);
}
// blockIdx.y picks the layer (or `one`), blockIdx.x walks its (row, k) pairs.  (Tried: a thread converting 8 consecutive k
// of a row for every tap and writing 16-byte pieces instead of 2-byte elements -- its strided reads made it 790 us per
// step against 340.)
__global__ __launch_bounds__(256) void wprep_kernel(const WprepDesc* __restrict__ descs, WprepDesc one) {
  const WprepDesc d = descs ? descs[blockIdx.y] : one;
  const long total = (long)d.Mpad * d.Kc;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float s = weight_scale(d.wp);
  unsigned short* out = reinterpret_cast<unsigned short*>(d.wp + HEADER);
  const int kl = (int)(i & 31);                      // k fastest inside a group of 32: contiguous 2-byte stores
  const long rest = i >> 5;
  const int row = (int)(rest % d.Mpad), cc = (int)(rest / d.Mpad), k = cc * 32 + kl;
  const long plane = (long)d.taps * d.Mpad * d.Kc;
  const float* src = d.w + (row < d.M ? row * d.sm + k * d.sk : 0);
  for (int t = 0; t < d.taps; ++t) {
    const float a = row < d.M ? s * src[d.flip ? d.taps - 1 - t : t] : 0.f;
    const _Float16 h0 = (_Float16)a, h1 = (_Float16)(a - (float)h0);
    const long o = (((long)cc * d.taps + t) * d.Mpad + row) * 32 + kl;
    out[o] = __builtin_bit_cast(unsigned short, h0);
    out[plane + o] = __builtin_bit_cast(unsigned short, h1);
  }
}
inline long wp_bytes(int M, int Kc, int taps) { return HEADER + 2L * taps * x3::pad128(M) * Kc * 2; }

// ---- loaders (same slicing contract as conv_x3.h: st(g) of a chunk precedes ld(g) of the next one) -------------------
// 16 fp32 values of one LDS row -> two fp16 planes.  Slices 0..7 convert one pair each, 8..11 write one 16-byte piece.
struct Split16 {
  unsigned pk[2][8];
  __device__ __forceinline__ void st(int g, const float (&r)[16], float s, unsigned char* tile, int row, int k0,
                                     unsigned okm = 0xffffu) {
    if (g < 8) {
      const float v0 = s * igemm::keep_if(r[2 * g], okm, 2 * g), v1 = s * igemm::keep_if(r[2 * g + 1], okm, 2 * g + 1);
      pk[0][g] = pack_h2(v0, v1);
      pk[1][g] = pack_h2(lo_part(v0), lo_part(v1));
    } else if (g < 12) {
      const int sp = (g - 8) >> 1, h = (g - 8) & 1;
      *reinterpret_cast<u32x4*>(tile + sp * PLANE + row * PITCH + k0 * 2 + h * 16) =
          u32x4{pk[sp][4 * h], pk[sp][4 * h + 1], pk[sp][4 * h + 2], pk[sp][4 * h + 3]};
    }
  }
};

struct WpLoader {        // four 16-byte pieces per chunk: split j>>1, row (tid>>2) + 64 (j&1), segment tid&3
  __amdgpu_buffer_rsrc_t rs; unsigned voff, plane2, chunk2; int row, seg;
  u32x4 r[4];
  __device__ __forceinline__ void init(const x3::Src& q, int m0, int tid) {
    row = tid >> 2; seg = tid & 3;
    plane2 = (unsigned)q.taps * q.Mpad * q.Kc * 2u;
    chunk2 = (unsigned)q.Mpad * KC * 2u;
    rs = x3::make_rsrc(reinterpret_cast<const unsigned char*>(q.wp) + HEADER, 2L * plane2);
    voff = ((unsigned)(m0 + row) * KC + seg * 8) * 2u;
  }
  __device__ __forceinline__ void ld(int g, int c) {
    if (g >= 4) return;
    r[g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                         rs, (int)voff, (int)((unsigned)c * chunk2 + (g >> 1) * plane2 + (g & 1) * 64u * KC * 2u), 0));
  }
  __device__ __forceinline__ void st(int g, unsigned char* t) {
    if (g < 4) *reinterpret_cast<u32x4*>(t + (g >> 1) * PLANE + (row + 64 * (g & 1)) * PITCH + seg * 16) = r[g];
  }
};

// activations, lanes along pixels; a thread holds 16 channels of one tap-shifted pixel (x3::ActLoader's addressing)
template <bool DUAL, int TAPS>
struct ActLoader {
  __amdgpu_buffer_rsrc_t rs1, rs2;
  int nl, kg, tb1, tb2; unsigned mask; float scale;
  float r[16]; Split16 sp;
  __device__ __forceinline__ void init(const ConvP& p, const x3::Src& q, int n0, int tid, float s) {
    nl = tid & 127;
    kg = __builtin_amdgcn_readfirstlane(tid >> 7);      // which 16 of the chunk's 32 channels
    rs1 = x3::make_rsrc(q.s1, (long)p.N * q.S1 * p.HW * 4);
    rs2 = x3::make_rsrc(q.s2, (long)p.N * (DUAL ? q.S2 : q.S1) * p.HW * 4);
    mask = 0; tb1 = 0; tb2 = 0; scale = s;
    const int n = n0 + nl;
    if (n < p.N * p.HW) {
      const int b = n / p.HW, hw = n - b * p.HW;
      const int y = hw / p.W, x = hw - y * p.W;
      if (TAPS == 1) mask = 1u;
#pragma unroll
      for (int t = 0; t < (TAPS == 9 ? 9 : 0); ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << t;
      }
      tb1 = b * q.S1 * p.HW + hw;
      tb2 = b * q.S2 * p.HW + hw;
    }
  }
  // slices 8..23 load one channel each (its register was consumed by conversion slice (g - 8) / 2 <= 7)
  __device__ __forceinline__ void ld(int g, const ConvP& p, const x3::Src& q, int c) {
    if (g < 8) return;
    const int cc = TAPS == 9 ? c / 9 : c, tap = c - cc * TAPS;   // scalar: chunk = (32-channel group, tap), tap fastest
    const int ci0 = cc * KC + kg * 16;
    const bool first = !DUAL || ci0 < q.S1;                      // scalar: a chunk never straddles the two sources
    const __amdgpu_buffer_rsrc_t rs = first ? rs1 : rs2;
    const unsigned so = (unsigned)(first ? ci0 : ci0 - q.S1) * p.HW * 4u;
    const unsigned dead = (((mask >> tap) & 1u) ^ 1u) << 31;     // halo lanes: outside the buffer -> 0
    const int shift = TAPS == 9 ? (tap / 3 - 1) * p.W + (tap % 3 - 1) : 0;
    const unsigned vo = (unsigned)(((first ? tb1 : tb2) + shift) * 4) | dead;
    r[g - 8] = x3::bload(rs, vo, so + (unsigned)(g - 8) * p.HW * 4u);
  }
  __device__ __forceinline__ void st(int g, unsigned char* t) { sp.st(g, r, scale, t, nl, kg * 16); }
};

// (Round 4, measured and dropped: a loader for the 1x1 layers that fetches 4 pixels x 4 channels per thread with 16-byte loads
// instead of 16 four-byte ones -- what had made the per-tap weight gradient 30 % faster -- changes nothing here: 384 -> 128 at
// 32 x 32 53.2 vs 52.3 us, 256 -> 256 data gradient 42.6 vs 41.3.  These kernels are not address-bound: a workgroup has 8-16
// chunks in all, and the chunk time is the staging round trip of the fp32-operand pipeline, profiles/r04_experiments.txt.)
// ---- the kernel: out tile 128 x 128, 4 waves of 64 x 64, chunks of 32 k; 12 MFMAs per 16-k step and wave -------------
// Grid (XCD-remapped): one flat dimension of tiles x K-splits.  xpart: `nxpart` (256 / 512) partial |x| maxima.
template <class BL, class EP, int WPS = 2>       // WPS: waves per SIMD the register allocation must admit
__global__ __launch_bounds__(256, WPS) void gemm_kernel(ConvP p, x3::Src q, int M, int Nn, int tiles_m, int tiles_n,
                                                   int nchunks_total, int chunks_per_split, const float* __restrict__ xpart,
                                                   int nxpart) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  unsigned char* As = lds;
  unsigned char* Bs = lds + OPER;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const float sx = pow2_scale_of(block_amax(xpart, nxpart, reinterpret_cast<float*>(lds)));
  const float sw = weight_scale(q.wp);
  const float unscale = 1.f / (sw * sx);                                   // a power of two: exact
  const int ntiles = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id % ntiles, zs = id / ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * 128;
  const int c_begin = zs * chunks_per_split;
  const int c_last = min(nchunks_total, c_begin + chunks_per_split) - 1;     // >= c_begin by construction

  WpLoader al; BL bl;
  al.init(q, m0, tid);
  bl.init(p, q, n0, tid, sx);
  EP ep;
  ep.preload(p, m0, n0, 128, M, Nn, tid);

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

  // Pipeline as in x3::gemm_kernel (LDS single-buffered, two barriers per chunk); with 12 MFMAs per half chunk each
  // MFMA of the second half carries TWO staging slices.
#define STK_X2_FRAGS(KK)                                                                               \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int s = 0; s < 2; ++s) {         \
    a[i][s] = *reinterpret_cast<const halfx8*>(a_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);          \
    b[i][s] = *reinterpret_cast<const halfx8*>(b_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);          \
  }
  // three products per tile, the two cross terms first; MFMA g: product g / 4, tile (g / 2) & 1, g & 1
  constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};
#define STK_X2_MFMA(G)                                                                                              \
  acc[((G) >> 1) & 1][(G) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[((G) >> 1) & 1][SA[(G) >> 2]], b[(G) & 1][SB[(G) >> 2]], \
                                                                        acc[((G) >> 1) & 1][(G) & 1], 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 24; ++g) { al.ld(g, c_begin); bl.ld(g, p, q, c_begin); }
  {
    const int c1 = min(c_begin + 1, c_last);
#pragma unroll
    for (int g = 0; g < 24; ++g) { al.st(g, As); bl.st(g, Bs); al.ld(g, c1); bl.ld(g, p, q, c1); }
  }
  halfx8 a[2][2], b[2][2];
  for (int c = c_begin; c < c_last; ++c) {
    __syncthreads();                                   // B1: chunk c is in LDS
    STK_X2_FRAGS(0)
#pragma unroll
    for (int g = 0; g < 12; ++g) { STK_X2_MFMA(g) }
    STK_X2_FRAGS(1)
    __syncthreads();                                   // B2: nobody reads LDS any more
    const int c2 = min(c + 2, c_last);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      STK_X2_MFMA(g)
      al.st(2 * g, As); bl.st(2 * g, Bs); al.st(2 * g + 1, As); bl.st(2 * g + 1, Bs);      // chunk c + 1 -> LDS
      al.ld(2 * g, c2); bl.ld(2 * g, p, q, c2); al.ld(2 * g + 1, c2); bl.ld(2 * g + 1, p, q, c2);   // chunk c + 2
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  STK_X2_FRAGS(0)
#pragma unroll
  for (int g = 0; g < 12; ++g) { STK_X2_MFMA(g) }
  STK_X2_FRAGS(1)
#pragma unroll
  for (int g = 0; g < 12; ++g) { STK_X2_MFMA(g) }
#undef STK_X2_MFMA
#undef STK_X2_FRAGS

  ep.stage(lds, tid);
  ep.init(p, 0, zs);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn0 + j * 32 + fc;
    const bool nok = n < Nn;
    ep.col(p, nok ? n : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] *= unscale;
      ep.strip(p, m0 + wm0 + i * 32 + 4 * fk, M, nok, nok ? n : 0, acc[i][j]);
    }
  }
}

// ---- 3x3 weight gradient, three taps per workgroup, on the two-way split ---------------------------------------------
// x3::wgrad3_kernel (same tiling, same loads, same slab layout) with both operands scaled and split into two fp16
// terms: 18 instead of 36 MFMAs per wave and half chunk.  The loaders are x3's with the splitter replaced.
constexpr int W3_BPLANE = 64 * PITCH;
constexpr int W3_BTAP = 2 * W3_BPLANE;
constexpr int W3_LDS = OPER + 3 * W3_BTAP;       // 20480 + 30720 = 51200 bytes

struct RowsA : x3::RowsLoader<false, false, 16> {          // dy: 16 consecutive pixels of one output channel
  Split16 sp2; float scale;
  __device__ __forceinline__ void st(int g, unsigned char* t) { sp2.st(g, r, scale, t, row, half * 16, okm); }
};

template <bool DUAL>
struct Rows3 : x3::Rows3Loader<DUAL> {                     // x: the 10-pixel run of one input channel, three windows
  using B = x3::Rows3Loader<DUAL>;
  _Float16 hh[2][10]; float scale;
  // slices 0..4 split two elements each, 5..10 pack + write one (tap, plane) window each
  __device__ __forceinline__ void st(int g, unsigned char* t) {
    if (g < 5) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = 2 * g + u;
        const float v = scale * igemm::keep_if(B::r[e], B::okm, e);
        hh[0][e] = (_Float16)v;
        hh[1][e] = (_Float16)(v - (float)hh[0][e]);
      }
    } else if (g < 11) {
      const int tap = (g - 5) >> 1, s = (g - 5) & 1;        // window of tap kw: elements kw - 1 .. kw + 6 -> hh[kw .. kw + 7]
      unsigned char* d = t + tap * W3_BTAP + s * W3_BPLANE + B::row * PITCH + B::quarter * 16;
      unsigned w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const halfx2 v = {hh[s][tap + 2 * j], hh[s][tap + 2 * j + 1]};
        w[j] = __builtin_bit_cast(unsigned, v);
      }
      *reinterpret_cast<u32x4*>(d) = u32x4{w[0], w[1], w[2], w[3]};
    }
  }
};

// dypart: 256 partial |dy| maxima; xpart: nxpart (256 / 512) partial |x| maxima
template <bool DUAL>
__global__ __launch_bounds__(256) void wgrad3_kernel(ConvP p, int tiles_m, int tiles_n, int nchunks_total, int chunks_per_split,
                                                     const float* __restrict__ dypart, const float* __restrict__ xpart, int nxpart) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[W3_LDS];
  unsigned char* As = lds;
  unsigned char* Bs = lds + OPER;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const float sa = pow2_scale_of(block_amax(dypart, NPART, reinterpret_cast<float*>(lds)));
  const float sb = pow2_scale_of(block_amax(xpart, nxpart, reinterpret_cast<float*>(lds)));
  const float unscale = 1.f / (sa * sb);
  const int ntiles = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);       // kernel row fastest: the three blocks share dy / x panels
  const int kh = id % 3;
  const int rest = id / 3;
  const int tile = rest % ntiles;
  const int zs = rest / ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * 64;
  const int c_begin = zs * chunks_per_split;
  const int c_last = min(nchunks_total, c_begin + chunks_per_split) - 1;

  x3::Src q = {};
  RowsA al;
  Rows3<DUAL> bl;
  al.init(p, q, m0, tid, 4); al.scale = sa;
  bl.init(p, n0, tid, kh); bl.scale = sb;

  floatx16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.f;

  const int wm0 = (wid & 1) * 64, wn0 = (wid >> 1) * 32;
  const int fk = lane >> 5, fc = lane & 31;
  const unsigned char* a_rd = As + (wm0 + fc) * PITCH + fk * 16;
  const unsigned char* b_rd = Bs + (wn0 + fc) * PITCH + fk * 16;

#define STK_W3_FRAGS(KK)                                                                                   \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
      a[i][s] = *reinterpret_cast<const halfx8*>(a_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);             \
    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                            \
      b[t][s] = *reinterpret_cast<const halfx8*>(b_rd + t * W3_BTAP + s * W3_BPLANE + (KK) * 32);            \
  }
  constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};
#define STK_W3_MFMAS                                                                                       \
  _Pragma("unroll") for (int pr = 0; pr < 3; ++pr) _Pragma("unroll") for (int i = 0; i < 2; ++i)              \
    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                            \
      acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][SA[pr]], b[t][SB[pr]], acc[i][t], 0, 0, 0);

#pragma unroll
  for (int g = 0; g < 24; ++g) { al.ld(g, p, q, c_begin); bl.ld(g, p, c_begin); }
  {
    const int c1 = min(c_begin + 1, c_last);
#pragma unroll
    for (int g = 0; g < 24; ++g) { al.st(g, As); bl.st(g, Bs); }
#pragma unroll
    for (int g = 0; g < 24; ++g) { al.ld(g, p, q, c1); bl.ld(g, p, c1); }
  }
  halfx8 a[2][2], b[3][2];
  for (int c = c_begin; c < c_last; ++c) {
    __syncthreads();                                   // chunk c is in LDS
    STK_W3_FRAGS(0)
    STK_W3_MFMAS
    STK_W3_FRAGS(1)
    __syncthreads();                                   // nobody reads LDS any more
    const int c2 = min(c + 2, c_last);
    STK_W3_MFMAS
#pragma unroll
    for (int g = 0; g < 24; ++g) al.st(g, As);
#pragma unroll
    for (int g = 0; g < 24; ++g) bl.st(g, Bs);
#pragma unroll
    for (int g = 0; g < 24; ++g) al.ld(g, p, q, c2);
#pragma unroll
    for (int g = 0; g < 24; ++g) bl.ld(g, p, c2);
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);     // VALU
      if (g < 10) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write (4 + 6 per thread and chunk)
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     // VMEM read
    }
  }
  __syncthreads();
  STK_W3_FRAGS(0)
  STK_W3_MFMAS
  STK_W3_FRAGS(1)
  STK_W3_MFMAS
#undef STK_W3_MFMAS
#undef STK_W3_FRAGS

  // partial slab of split zs as [tap][Cout][Cin] (lanes = ci, contiguous); splitk_reduce_kernel re-lays it out
  float* slab = p.part + (long)zs * p.part_stride;
  const int n = n0 + wn0 + fc;
  if (n < p.Cin) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm0 + i * 32 + 4 * fk + igemm::strip_row(e);
          if (m < p.Cout) slab[((long)(kh * 3 + t) * p.Cout + m) * p.Cin + n] = unscale * acc[i][t][e];
        }
  }
}

// ---- per-tap weight gradient (1x1 layers, 3x3 on 4-wide maps) on the two-way split --------------------------------------
// x3::gemm_kernel<RowsLoader, RowsLoader, EpWgrad> with both operands scaled and split into two fp16 terms: 12 instead
// of 24 MFMAs per 16-k step and ~6 instead of ~10 conversion VALU per element.  With half the MFMAs the conversions no
// longer hide behind the second half-chunk alone, so a chunk's work is laid out over both halves: the split of chunk
// c + 1 (registers only) and the load issue of chunk c + 2 ride on the MFMAs of kk = 0, the eight LDS stores on those of
// kk = 1 (after the barrier that retires the reads of chunk c).
template <bool DUAL, int SEG>
struct RowsB : x3::RowsLoader<true, DUAL, SEG> {           // x: 16 consecutive (tap-shifted) pixels of one input channel
  using B = x3::RowsLoader<true, DUAL, SEG>;
  Split16 sp2; float scale;
  __device__ __forceinline__ void st(int g, unsigned char* t) { sp2.st(g, B::r, scale, t, B::row, B::half * 16, B::okm); }
};

// Unshifted rows (dy always; x of a 1x1 layer): the thread's 16 consecutive pixels are one aligned 64-byte run, read as
// four 16-byte loads.  (x3::RowsLoader reads a run element by element because a tap shift breaks the alignment and needs
// a per-element mask; with 32 such loads per thread and chunk, each touching 64 separate 64-byte segments, the per-tap
// kernel was bound by address processing: 54 us for the 256 -> 256 layer at 16 x 16 against ~6 us of MFMAs.)
template <bool IS_X, bool DUAL>
struct RowsU {
  __amdgpu_buffer_rsrc_t rs;
  int rowoff, bstride, row, half; bool rowok; unsigned voff;
  float r[16]; Split16 sp2; float scale;
  __device__ __forceinline__ void init(const ConvP& p, const x3::Src&, int o0, int tid, int) {
    row = tid >> 1; half = tid & 1; voff = 0x80000000u;
    const int ch = o0 + row;
    if (IS_X) {
      rowok = ch < p.Cin;
      const int c = rowok ? ch : 0;
      const bool first = !DUAL || __builtin_amdgcn_readfirstlane(o0 + (tid >> 6) * 32) < p.C1;   // wave-uniform
      rs = first ? x3::make_rsrc(p.x1, (long)p.N * p.C1 * p.HW * 4) : x3::make_rsrc(p.x2, (long)p.N * p.C2 * p.HW * 4);
      rowoff = (first ? c : (c >= p.C1 ? c - p.C1 : 0)) * p.HW;
      bstride = (first ? p.C1 : p.C2) * p.HW;
    } else {
      rowok = ch < p.Cout;
      rs = x3::make_rsrc(p.dy, (long)p.N * p.Cout * p.HW * 4);
      rowoff = (rowok ? ch : 0) * p.HW;
      bstride = p.Cout * p.HW;
    }
  }
  // slice 14: the run's byte offset (bit 31 = outside: the loads return 0); slices 15..18: one 16-byte load each
  __device__ __forceinline__ void ld(int g, const ConvP& p, const x3::Src&, int c) {
    if (g == 14) {
      const int k = c * KC + half * 16;
      const bool kin = rowok && k < p.N * p.HW;
      const int b = k >> p.ohw_shift, hw = k & (p.HW - 1);
      voff = kin ? (unsigned)((b * bstride + rowoff + hw) * 4) : 0x80000000u;
    } else if (g >= 15 && g < 19) {
      const int j = g - 15;
      const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, j * 16, 0));
      r[4 * j] = __uint_as_float(v[0]); r[4 * j + 1] = __uint_as_float(v[1]);
      r[4 * j + 2] = __uint_as_float(v[2]); r[4 * j + 3] = __uint_as_float(v[3]);
    }
  }
  __device__ __forceinline__ void st(int g, unsigned char* t) { sp2.st(g, r, scale, t, row, half * 16); }
};

template <class AL, class BL, class EP, bool PIN>
__global__ __launch_bounds__(256) void wgemm_kernel(ConvP p, int M, int Nn, int tiles_m, int tiles_n, int nchunks_total,
                                                    int chunks_per_split, int taps_z, const float* __restrict__ dypart,
                                                    const float* __restrict__ xpart, int nxpart) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  unsigned char* As = lds;
  unsigned char* Bs = lds + OPER;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const float sa = pow2_scale_of(block_amax(dypart, NPART, reinterpret_cast<float*>(lds)));
  const float sb = pow2_scale_of(block_amax(xpart, nxpart, reinterpret_cast<float*>(lds)));
  const float unscale = 1.f / (sa * sb);
  const int ntiles = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int zb = id % taps_z;
  const int rest = id / taps_z;
  const int tile = rest % ntiles;
  const int zs = rest / ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * 128;
  const int c_begin = zs * chunks_per_split;
  const int c_last = min(nchunks_total, c_begin + chunks_per_split) - 1;

  x3::Src q = {};
  AL al; BL bl;
  al.init(p, q, m0, tid, 4); al.scale = sa;
  bl.init(p, q, n0, tid, taps_z == 1 ? 4 : zb); bl.scale = sb;

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

#define STK_W1_FRAGS(KK)                                                                               \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int s = 0; s < 2; ++s) {         \
    a[i][s] = *reinterpret_cast<const halfx8*>(a_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);          \
    b[i][s] = *reinterpret_cast<const halfx8*>(b_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);          \
  }
  constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};
#define STK_W1_MFMA(G)                                                                                              \
  acc[((G) >> 1) & 1][(G) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[((G) >> 1) & 1][SA[(G) >> 2]], b[(G) & 1][SB[(G) >> 2]], \
                                                                        acc[((G) >> 1) & 1][(G) & 1], 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 24; ++g) { al.ld(g, p, q, c_begin); bl.ld(g, p, q, c_begin); }
#pragma unroll
  for (int g = 0; g < 12; ++g) { al.st(g, As); bl.st(g, Bs); }
  {
    const int c1 = min(c_begin + 1, c_last);
#pragma unroll
    for (int g = 0; g < 24; ++g) { al.ld(g, p, q, c1); bl.ld(g, p, q, c1); }
  }
  halfx8 a[2][2], b[2][2];
  for (int c = c_begin; c < c_last; ++c) {
    __syncthreads();                                   // chunk c is in LDS
    STK_W1_FRAGS(0)
#pragma unroll
    for (int g = 0; g < 12; ++g) { STK_W1_MFMA(g) }
#pragma unroll
    for (int g = 0; g < 8; ++g) { al.st(g, As); bl.st(g, Bs); }            // split of chunk c + 1: registers only
    const int c2 = min(c + 2, c_last);
#pragma unroll
    for (int g = 0; g < 24; ++g) { al.ld(g, p, q, c2); bl.ld(g, p, q, c2); }   // chunk c + 2: global -> registers
    if (PIN) {
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 18, 0);    // VALU
        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);     // VMEM read
      }
    }
    STK_W1_FRAGS(1)
    __syncthreads();                                   // nobody reads chunk c any more
#pragma unroll
    for (int g = 0; g < 12; ++g) { STK_W1_MFMA(g) }
#pragma unroll
    for (int g = 8; g < 12; ++g) { al.st(g, As); bl.st(g, Bs); }           // chunk c + 1 -> LDS
    if (PIN) {
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (g >= 2 && g < 10) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
      }
    }
  }
  __syncthreads();
  STK_W1_FRAGS(0)
#pragma unroll
  for (int g = 0; g < 12; ++g) { STK_W1_MFMA(g) }
  STK_W1_FRAGS(1)
#pragma unroll
  for (int g = 0; g < 12; ++g) { STK_W1_MFMA(g) }
#undef STK_W1_MFMA
#undef STK_W1_FRAGS

  EP ep;
  ep.init(p, zb, zs);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn0 + j * 32 + fc;
    const bool nok = n < Nn;
    ep.col(p, nok ? n : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] *= unscale;
      ep.strip(p, m0 + wm0 + i * 32 + 4 * fk, M, nok, nok ? n : 0, acc[i][j]);
    }
  }
}

}  // namespace x2
