// upfirdn2d.hip -- FIR up/down-sampling for gfx950 (stk_upfirdn2d_f32 / stk_upfirdn2d_acc_f32).
//
// Replaces op/upfirdn2d_kernel.cu of the reference (its tiled template kernel :107-207 and the
// generic fallback :50-105).  Semantics (op/upfirdn2d.py:159-200): zero-upsample by `up`
// (zeros appended after each sample), pad/crop, true convolution with the taps, keep every
// `down`-th sample:
//
//   out[oy,ox] = sum_{ky,kx} k[kh-1-ky, kw-1-kx] * in[(oy*down+ky-pad_y0)/up, (ox*down+kx-pad_x0)/up]
//                over the taps whose upsampled coordinate is >= 0, divisible by `up` and in range.
//
// The op is HBM-bound (algorithmic bytes = in + out, 4x4 taps).  Design for CDNA4:
//  * one 256-thread workgroup produces a tile of 1024 outputs of one plane: TOW = min(64, pow2(out_w))
//    columns x 1024/TOW rows; the input window of the tile (<= 5.4K floats) is staged in LDS with
//    coalesced row reads and zero fill, so every input element leaves HBM once and the 16 (down) or
//    4 (up) taps per output are LDS reads;
//  * lanes map to consecutive output columns -> coalesced 256 B stores per wave, and each thread
//    keeps 4 output rows in registers so the taps are read from LDS once per column offset;
//  * up and down are template constants (1 or 2) so the divisibility tests and divisions fold away;
//  * tiny planes (fewer than 256 outputs) and unusual factors use the direct kernel, whose reads are
//    served by L1/L2 -- there is too little data per plane for staging to pay.
#include "common.h"

namespace {

struct UfdParams {
  const float* in;
  const float* k;
  float* out;
  float beta;
  int major, in_h, in_w, minor, kh, kw;
  int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  int out_h, out_w;
};

__device__ __forceinline__ int floor_div(int a, int b) {
  int c = a / b;
  return (c * b > a) ? c - 1 : c;
}

// ---- direct kernel: one thread per output element, any parameters -------------------------------
__global__ __launch_bounds__(256) void upfirdn2d_direct(UfdParams p) {
  const long total = (long)p.major * p.out_h * p.out_w * p.minor;
  const long stride = (long)gridDim.x * 256;
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
    long r = o;
    const int mi = (int)(r % p.minor); r /= p.minor;
    const int ox = (int)(r % p.out_w); r /= p.out_w;
    const int oy = (int)(r % p.out_h);
    const int mj = (int)(r / p.out_h);
    float acc = 0.f;
    for (int ky = 0; ky < p.kh; ++ky) {
      const int uy = oy * p.down_y + ky - p.pad_y0;
      if (uy < 0 || uy % p.up_y) continue;
      const int iy = uy / p.up_y;
      if (iy >= p.in_h) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ux = ox * p.down_x + kx - p.pad_x0;
        if (ux < 0 || ux % p.up_x) continue;
        const int ix = ux / p.up_x;
        if (ix >= p.in_w) continue;
        acc += p.in[((long)(mj * p.in_h + iy) * p.in_w + ix) * p.minor + mi] *
               p.k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
      }
    }
    p.out[o] = (p.beta != 0.f ? p.beta * p.out[o] : 0.f) + acc;
  }
}

// ---- tiled kernel: minor == 1, up_x == up_y == UP, down_x == down_y == DOWN, kh,kw <= 8 -----------
constexpr int UFD_LDS_FLOATS = 5632;   // >= max over tile shapes of rows*(cols+1) for 8 taps, down 2
constexpr int UFD_MAX_TAPS = 8;

template <int UP, int DOWN>
__global__ __launch_bounds__(256) void upfirdn2d_tiled(UfdParams p, int tow_log2, int tiles_x, int tiles_y) {
  __shared__ float s_in[UFD_LDS_FLOATS];
  __shared__ float s_k[UFD_MAX_TAPS * UFD_MAX_TAPS];

  const int TOW = 1 << tow_log2;
  const int TOH = 1024 >> tow_log2;
  const int rows_per_pass = 256 >> tow_log2;      // thread rows per pass; 4 passes cover TOH

  int tile = blockIdx.x;
  const int tx_tile = tile % tiles_x; tile /= tiles_x;
  const int ty_tile = tile % tiles_y;
  const int plane = tile / tiles_y;
  const int oy0 = ty_tile * TOH, ox0 = tx_tile * TOW;

  // taps, flipped once: s_k[ky][kx] = k[kh-1-ky][kw-1-kx]
  if (threadIdx.x < p.kh * p.kw) {
    const int ky = threadIdx.x / p.kw, kx = threadIdx.x % p.kw;
    s_k[ky * UFD_MAX_TAPS + kx] = p.k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }

  // input window of this tile, in input coordinates (may start negative / end past the image)
  const int iy_lo = floor_div(oy0 * DOWN - p.pad_y0, UP);
  const int iy_hi = floor_div((oy0 + TOH - 1) * DOWN + p.kh - 1 - p.pad_y0, UP);
  const int ix_lo = floor_div(ox0 * DOWN - p.pad_x0, UP);
  const int ix_hi = floor_div((ox0 + TOW - 1) * DOWN + p.kw - 1 - p.pad_x0, UP);
  const int rows = iy_hi - iy_lo + 1, cols = ix_hi - ix_lo + 1;
  const int pitch = cols | 1;   // odd pitch: column-strided reads of the down-2 case stay 2-way at worst

  const float* src = p.in + (long)plane * p.in_h * p.in_w;
  for (int e = threadIdx.x; e < rows * cols; e += 256) {
    const int r = e / cols, c = e - r * cols;
    const int iy = iy_lo + r, ix = ix_lo + c;
    float v = 0.f;
    if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v = src[(long)iy * p.in_w + ix];
    s_in[r * pitch + c] = v;
  }
  __syncthreads();

  const int tx = threadIdx.x & (TOW - 1);
  const int ty = threadIdx.x >> tow_log2;
  const int ox = ox0 + tx;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int rbase[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) rbase[q] = (oy0 + ty + q * rows_per_pass) * DOWN - p.pad_y0;
  const int ux0 = ox * DOWN - p.pad_x0;

  for (int ky = 0; ky < p.kh; ++ky) {
    for (int kx = 0; kx < p.kw; ++kx) {
      const int ux = ux0 + kx;
      if (UP > 1 && (ux & (UP - 1))) continue;          // lands on an inserted zero
      const int c = floor_div(ux, UP) - ix_lo;
      const float w = s_k[ky * UFD_MAX_TAPS + kx];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int uy = rbase[q] + ky;
        if (UP > 1 && (uy & (UP - 1))) continue;
        const int r = floor_div(uy, UP) - iy_lo;
        acc[q] += s_in[r * pitch + c] * w;
      }
    }
  }

  if (ox < p.out_w) {
    float* dst = p.out + (long)plane * p.out_h * p.out_w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = oy0 + ty + q * rows_per_pass;
      if (oy < p.out_h) {
        float* o = dst + (long)oy * p.out_w + ox;
        *o = (p.beta != 0.f ? p.beta * *o : 0.f) + acc[q];
      }
    }
  }
}

int launch(const float* input, const float* kernel, float* out, float beta, int major, int in_h, int in_w, int minor,
           int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
           int pad_y1, hipStream_t stream) {
  if (!input || !kernel || !out || major <= 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0 ||
      up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0)
    return STK_EINVAL;
  UfdParams p;
  p.in = input; p.k = kernel; p.out = out; p.beta = beta;
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  if (p.out_h <= 0 || p.out_w <= 0) return STK_EINVAL;

  const bool tiled_ok = minor == 1 && up_x == up_y && down_x == down_y && kh <= UFD_MAX_TAPS && kw <= UFD_MAX_TAPS &&
                        ((up_x == 1 && (down_x == 1 || down_x == 2)) || (up_x == 2 && down_x == 1)) &&
                        (long)p.out_h * p.out_w >= 256;
  if (tiled_ok) {
    int tow_log2 = 4;                       // TOW in {16, 32, 64}
    while (tow_log2 < 6 && (1 << tow_log2) < p.out_w) ++tow_log2;
    const int TOW = 1 << tow_log2, TOH = 1024 >> tow_log2;
    const int tiles_x = stk_cdiv(p.out_w, TOW), tiles_y = stk_cdiv(p.out_h, TOH);
    const long nblk = (long)major * tiles_x * tiles_y;
    if (nblk <= 0x7fffffffL) {
      dim3 grid((unsigned)nblk), block(256);
      if (up_x == 1 && down_x == 1)
        hipLaunchKernelGGL((upfirdn2d_tiled<1, 1>), grid, block, 0, stream, p, tow_log2, tiles_x, tiles_y);
      else if (up_x == 1)
        hipLaunchKernelGGL((upfirdn2d_tiled<1, 2>), grid, block, 0, stream, p, tow_log2, tiles_x, tiles_y);
      else
        hipLaunchKernelGGL((upfirdn2d_tiled<2, 1>), grid, block, 0, stream, p, tow_log2, tiles_x, tiles_y);
      STK_CHECK_LAUNCH();
      return STK_OK;
    }
  }
  const long total = (long)major * p.out_h * p.out_w * minor;
  hipLaunchKernelGGL(upfirdn2d_direct, dim3(stk_ew_grid(total)), dim3(256), 0, stream, p);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

}  // namespace

extern "C" {

int stk_upfirdn2d_f32(const float* input, const float* kernel, float* out, int major, int in_h, int in_w, int minor,
                      int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                      int pad_y0, int pad_y1, void* stream) {
  return launch(input, kernel, out, 0.f, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0,
                pad_x1, pad_y0, pad_y1, (hipStream_t)stream);
}

int stk_upfirdn2d_acc_f32(const float* input, const float* kernel, float* out, float beta, int major, int in_h,
                          int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                          int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream) {
  return launch(input, kernel, out, beta, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0,
                pad_x1, pad_y0, pad_y1, (hipStream_t)stream);
}

}  // extern "C"
