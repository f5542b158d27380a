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

}  // namespace pl
