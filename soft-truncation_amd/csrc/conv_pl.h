// conv_pl.h -- "planes": activations pre-split for the fp16 two-way-split convolution kernels (conv_x2.h), for gfx950.
// Included by conv.hip inside its anonymous namespace, after conv_x2.h.
//
// Why.  x2::gemm_kernel computes an fp32 convolution as three fp16 MFMAs per product from operands split into two
// fp16 terms.  With fp32 NCHW activations its B loader does that split on the fly: per element mul, cvt, cvt, sub, cvt,
// pack -- and for a 3x3 layer nine times, once per tap.  That conversion VALU (about 100 issue slots per thread and
// 32-k chunk, against the 24 MFMAs of the chunk) is what kept the matrix pipe at 1.6 of 4 SIMDs busy
// (profiles/r01_traffic.json).  Here the split happens ONCE per tensor, and into the layout the consumer wants:
//
//   planes(N, C, HW):  P[split s][n][cb = c / 32][pixel][c % 32]  fp16 of  scale * x[n, c, pixel]
//       hi = fp16(scale x),  lo = fp16(scale x - hi);   C padded to a multiple of 32 with zeros;
//       one pixel's 32 channels of one plane are 64 contiguous bytes.
//
// so the B tile of a chunk (one tap x 32 channels x 128 pixels) is, per plane, 128 rows of 64 contiguous bytes at a
// tap-dependent pixel shift: the B loader becomes the same four 16-byte copies per thread as the weight loader, with
// the halo zeros still coming from the buffer range check (offset bit 31).
//
// scale is the power of two that puts the tensor's |x| bound in [2^13, 2^14) (x2::pow2_scale_of): the bound is either
// the measured maximum (256 partial maxima, as for the fp32 path) or an a-priori one (GroupNorm outputs:
// gn_bound_kernel in groupnorm.hip), stored in the same 256-float "amax" record that the consumer reads.
#pragma once

namespace pl {

using x2::HEADER;
using x2::NPART;
using x3::KC;
using x3::PITCH;
using x3::PLANE;

inline long plane_bytes(int N, int C, int HW) { return (long)N * ((C + 31) / 32) * HW * 64; }

// ---- fp32 NCHW -> planes ------------------------------------------------------------------------------------------
// One workgroup per (n, cb, 128-pixel tile): the 32 x 128 fp32 tile goes through LDS (pitch 129 floats: the float4
// loads of a wave are contiguous along pixels, the 8-channel gathers of the write phase hit 32 distinct banks), every
// thread then converts two 8-channel pieces and writes them as 16-byte stores, a wave covering 1 KB of each plane.
constexpr int SP_PIX = 128;
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int N, int C, int HW,
                                                           const float* __restrict__ amax, int namax,
                                                           unsigned char* __restrict__ out, long plane_stride) {
  __shared__ float t[32 * 129];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int tiles = (HW + SP_PIX - 1) / SP_PIX;
  const int Cb = (C + 31) >> 5;
  const int tile = blockIdx.x % tiles, rest = blockIdx.x / tiles;
  const int cb = rest % Cb, n = rest / Cb;
  const int p0 = tile * SP_PIX;
  const bool vec = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 256 * k;
    const int c = i >> 5, q4 = i & 31;
    const int ch = cb * 32 + c, px = p0 + 4 * q4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < C) {
      const float* src = x + ((long)n * C + ch) * HW + px;
      if (vec && px + 3 < HW) v = *reinterpret_cast<const float4*>(src);
      else {
        if (px < HW) v.x = src[0];
        if (px + 1 < HW) v.y = src[1];
        if (px + 2 < HW) v.z = src[2];
        if (px + 3 < HW) v.w = src[3];
      }
    }
    float* d = t + c * 129 + 4 * q4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  // scale from the amax record (every workgroup reduces it itself: no finishing launch); taken AFTER the tile loads were
  // issued so that the record's latency hides under them
  float m = 0.f;
  for (int i = tid; i < namax; i += 256) m = fmaxf(m, amax[i]);
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  const float s = x2::pow2_scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int id = tid + 256 * r;
    const int px = id >> 2, q = id & 3;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = s * t[(8 * q + 2 * j) * 129 + px], v1 = s * t[(8 * q + 2 * j + 1) * 129 + px];
      hi[j] = x2::pack_h2(v0, v1);
      lo[j] = x2::pack_h2(x2::lo_part(v0), x2::lo_part(v1));
    }
    if (p0 + px < HW) {
      unsigned char* o = out + (((long)n * Cb + cb) * HW + p0 + px) * 64 + q * 16;
      *reinterpret_cast<u32x4*>(o) = u32x4{hi[0], hi[1], hi[2], hi[3]};
      *reinterpret_cast<u32x4*>(o + plane_stride) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
  }
}

// ---- B loader of x2::gemm_kernel reading planes: four 16-byte copies per thread and chunk ---------------------------
// piece g: split g >> 1, row (tid >> 2) + 64 (g & 1), 16-byte segment tid & 3 -- x2::WpLoader's mapping.
template <int TAPS>
struct PlaneLoader {
  __amdgpu_buffer_rsrc_t rs; unsigned base[2], mask[2], ps; int row, seg;
  u32x4 r[4];
  __device__ __forceinline__ void init(const ConvP& p, const x3::Src& q, int n0, int tid, float) {
    row = tid >> 2; seg = tid & 3;
    ps = (unsigned)q.pl_stride;
    rs = x3::make_rsrc(q.pl, 2L * q.pl_stride);
    const int Cb = q.Kc >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + row + 64 * j;
      mask[j] = 0; base[j] = 0;
      if (n < p.N * p.HW) {
        const int b = n / p.HW, hw = n - b * p.HW;
        const int y = hw / p.W, x = hw - y * p.W;
        if (TAPS == 1) mask[j] = 1u;
#pragma unroll
        for (int t = 0; t < (TAPS == 9 ? 9 : 0); ++t) {
          const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
          if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask[j] |= 1u << t;
        }
        base[j] = ((unsigned)(b * Cb) * p.HW + hw) * 64u + seg * 16u;
      }
    }
  }
  __device__ __forceinline__ void ld(int g, const ConvP& p, const x3::Src&, int c) {
    if (g >= 4) return;
    const int cc = TAPS == 9 ? c / 9 : c, tap = c - cc * TAPS;   // scalar
    const int j = g & 1;
    const int shift = TAPS == 9 ? ((tap / 3 - 1) * p.W + (tap % 3 - 1)) * 64 : 0;
    const unsigned dead = (((mask[j] >> tap) & 1u) ^ 1u) << 31;  // halo rows: outside the buffer -> 0
    const unsigned vo = (base[j] + (unsigned)shift) | dead;
    r[g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                         rs, (int)vo, (int)((unsigned)cc * (unsigned)p.HW * 64u + (g >> 1) * ps), 0));
  }
  __device__ __forceinline__ void st(int g, unsigned char* t) {
    if (g < 4) *reinterpret_cast<u32x4*>(t + (g >> 1) * PLANE + (row + 64 * (g & 1)) * PITCH + seg * 16) = r[g];
  }
};

}  // namespace pl

// ---- double-buffered variant of x2::gemm_kernel for plane operands ---------------------------------------------------
// Both operands are 16-byte copies now, so the staging of a chunk is 8 loads + 8 ds_write_b128 per thread with no
// conversion in between.  Two LDS buffers (80 KB per workgroup, two workgroups per CU = all 160 KB) take the second
// barrier out of the chunk loop: while the waves read chunk c from one buffer, chunk c + 1 (already in registers) is
// written to the other and chunk c + 2 is requested.  One barrier per chunk; staging slices ride behind every MFMA
// of the chunk instead of only behind its second half.
namespace pl {

constexpr int DB_LDS = 2 * x2::LDS_BYTES;      // 81920

// ABL (ablation, benchmarks only -- results are garbage unless 0): bit 0 drops the LDS writes of the staging, bit 1 its
// global loads, bit 2 the per-chunk operand reads from LDS (the MFMAs then reuse the first chunk's registers).
template <class BL, class EP, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_db_kernel(ConvP p, x3::Src q, int M, int Nn, int tiles_m, int tiles_n,
                                                         int nchunks_total, int chunks_per_split,
                                                         const float* __restrict__ xpart, int nxpart) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[DB_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const float sx = x2::pow2_scale_of(x2::block_amax(xpart, nxpart, reinterpret_cast<float*>(lds)));
  const float sw = x2::weight_scale(q.wp);
  const float unscale = 1.f / (sw * sx);
  const int ntiles = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id % ntiles, zs = id / ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * 128;
  const int c_begin = zs * chunks_per_split;
  const int c_last = min(nchunks_total, c_begin + chunks_per_split) - 1;

  x2::WpLoader al; BL bl;
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
  const int a_off = (wm0 + fc) * PITCH + fk * 16;
  const int b_off = x2::OPER + (wn0 + fc) * PITCH + fk * 16;

#define STK_DB_FRAGS(BUF, KK)                                                                           \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int s = 0; s < 2; ++s) {          \
    a[i][s] = *reinterpret_cast<const halfx8*>((BUF) + a_off + s * PLANE + i * 32 * PITCH + (KK) * 32);  \
    b[i][s] = *reinterpret_cast<const halfx8*>((BUF) + b_off + s * PLANE + i * 32 * PITCH + (KK) * 32);  \
  }
  constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};
#define STK_DB_MFMA(G)                                                                                              \
  acc[((G) >> 1) & 1][(G) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[((G) >> 1) & 1][SA[(G) >> 2]], b[(G) & 1][SB[(G) >> 2]], \
                                                                        acc[((G) >> 1) & 1][(G) & 1], 0, 0, 0);
  // prologue: chunk c_begin -> buffer 0, chunk c_begin + 1 -> registers
#pragma unroll
  for (int g = 0; g < 4; ++g) { al.ld(g, c_begin); bl.ld(g, p, q, c_begin); }
  {
    const int c1 = min(c_begin + 1, c_last);
#pragma unroll
    for (int g = 0; g < 4; ++g) { al.st(g, lds); bl.st(g, lds + x2::OPER); al.ld(g, c1); bl.ld(g, p, q, c1); }
  }
  halfx8 a[2][2], b[2][2];
  int cur = 0;
  for (int c = c_begin; c < c_last; ++c) {
    unsigned char* rd = lds + cur * x2::LDS_BYTES;
    unsigned char* wr = lds + (cur ^ 1) * x2::LDS_BYTES;
    __syncthreads();                                   // chunk c is in `rd`; nobody reads `wr` (chunk c - 1) any more
    if (!(ABL & 4) || c == c_begin) { STK_DB_FRAGS(rd, 0) }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      STK_DB_MFMA(g)
      if (!(ABL & 1) && g >= 2 && g < 6) { al.st(g - 2, wr); bl.st(g - 2, wr + x2::OPER); }      // chunk c + 1 -> the other buffer
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(ABL & 4)) { STK_DB_FRAGS(rd, 1) }
    const int c2 = min(c + 2, c_last);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      STK_DB_MFMA(g)
      if (!(ABL & 2) && g >= 2 && g < 6) { al.ld(g - 2, c2); bl.ld(g - 2, p, q, c2); }           // chunk c + 2 -> registers
      __builtin_amdgcn_sched_barrier(0);
    }
    cur ^= 1;
  }
  {
    unsigned char* rd = lds + cur * x2::LDS_BYTES;
    __syncthreads();
    STK_DB_FRAGS(rd, 0)
#pragma unroll
    for (int g = 0; g < 12; ++g) { STK_DB_MFMA(g) }
    STK_DB_FRAGS(rd, 1)
#pragma unroll
    for (int g = 0; g < 12; ++g) { STK_DB_MFMA(g) }
  }
#undef STK_DB_MFMA
#undef STK_DB_FRAGS

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

// Kernel for plane operands, STK_PL_KERNEL (A/B switch of the kernel benchmarks; all give the same results up to the
// accumulation order): 4 (default) = LDS-DMA staging, 128 x 128 tiles (conv_x2d.h); 3 = the same with 128 x 256 tiles
// where they fill the chip; 1 = register staging, double-buffered LDS (above); 0 / 2 = x2::gemm_kernel's structure with
// the plane loader (2 / 3 waves per SIMD); >= 16 = ablation builds of the double-buffered kernel.
inline int kernel_choice() {
  static const int v = [] { const char* e = getenv("STK_PL_KERNEL"); return e ? atoi(e) : 4; }();
  return v;
}

}  // namespace pl
