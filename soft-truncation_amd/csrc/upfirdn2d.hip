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
//  * one 256-thread workgroup produces 1024-4096 outputs: a TOH x TOW tile (powers of two) of one plane, or -- for
//    the small feature maps, where a whole plane is only 16-256 outputs -- the whole plane of PPB consecutive
//    planes (one contiguous run of memory, read as float4); the input windows (<= 48 KB, LDS sized per launch)
//    are staged in LDS with zero fill, so every input element leaves HBM once and the 16 (down) or 4 (up) taps
//    per output are LDS reads;
//  * lanes map to consecutive output columns -> coalesced stores; all index decodes are shifts;
//  * up and down are template constants (1 or 2) so the divisibility tests and divisions fold away;
//  * unusual factors / minor > 1 use the direct kernel, whose reads are served by L1/L2.
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

// ---- tiled kernel: minor == 1, up_x == up_y == UP, down_x == down_y == DOWN, kh,kw <= 8 -----------------
// A 256-thread workgroup produces 256*NQ outputs = PPB planes x (TOH x TOW) tile, all powers of two so every
// output index decode is a shift.  Large planes: PPB = 1 and a grid of tiles per plane.  Small planes (the
// 16x16 / 8x8 feature maps, where one plane is only 64-256 outputs): the tile is the whole plane and one
// workgroup takes PPB consecutive planes -- which are one contiguous run of memory, read with float4 lanes.
// LDS is sized per launch (windows + taps) so the small-window cases keep 8 workgroups per CU.
constexpr int UFD_MAX_TAPS = 8;
constexpr int UFD_LDS_FLOATS = 12288;   // at most 48 KB of input windows per workgroup

// Row pitch of a staged window of `cols` columns.  Plain / up-sampling: odd.  Down-sampling by 2: a window row is kept
// de-interleaved -- its even columns, then (half a pitch on, a multiple of 16 bytes) its odd columns -- because an output
// column reads window columns 2 ox .. 2 ox + 3: with interleaved rows the lanes of a wave, four outputs each, read single
// floats 32 bytes apart (8 of the 64 banks; 4-way conflicts as soon as a wave spans 32 column groups, i.e. on rows of 128
// outputs), de-interleaved they read 16-byte pieces that are contiguous from lane to lane.
// (Rows of up to 64 outputs keep the interleaved form with an odd pitch: a wave then spans eight or more rows whose bank
// offsets differ, the single-float reads are conflict-free and measured faster -- 57.8 vs 64.9 us on 16384 planes 64 -> 32.)
__host__ __device__ inline bool ufd_deint(int down, int tow_log2) { return down == 2 && tow_log2 >= 7; }
__host__ __device__ inline int ufd_pitch(int cols, bool deint) {
  return deint ? 2 * ((((cols + 1) >> 1) + 3) & ~3) : (cols | 1);
}
__device__ __forceinline__ int ufd_col(int c, int pitch, bool deint) {
  return deint ? ((c & 1) ? (pitch >> 1) : 0) + (c >> 1) : c;
}

template <int UP, int DOWN, bool DEINT>
__global__ __launch_bounds__(256) void upfirdn2d_tiled(UfdParams p, int tow_log2, int toh_log2, int ppb_log2,
                                                       int nq, int tiles_x, int tiles_y, int whole, int vec, int k4) {
  extern __shared__ __attribute__((aligned(16))) float s_ufd[];
  float* s_k = s_ufd;                                  // [8][8] flipped taps
  float* s_in = s_ufd + UFD_MAX_TAPS * UFD_MAX_TAPS;   // PPB windows of rows x pitch

  const int TOW = 1 << tow_log2, TOH = 1 << toh_log2, PPB = 1 << ppb_log2;
  int tile = blockIdx.x;
  const int tx_tile = tile % tiles_x; tile /= tiles_x;
  const int ty_tile = tile % tiles_y;
  const int plane0 = (tile / tiles_y) << ppb_log2;
  const int oy0 = ty_tile * TOH, ox0 = tx_tile * TOW;

  // taps, flipped once: s_k[ky][kx] = k[kh-1-ky][kw-1-kx]; the load is issued now, stored after the staging
  // loads are in flight
  float kv = 0.f;
  const int tky = threadIdx.x / p.kw, tkx = threadIdx.x % p.kw;
  if (threadIdx.x < p.kh * p.kw) kv = p.k[(p.kh - 1 - tky) * p.kw + (p.kw - 1 - tkx)];

  // input window of the tile, in input coordinates (may start negative / end past the image)
  const int iy_lo = floor_div(oy0 * DOWN - p.pad_y0, UP);
  const int iy_hi = floor_div((oy0 + TOH - 1) * DOWN + p.kh - 1 - p.pad_y0, UP);
  const int ix_lo = floor_div(ox0 * DOWN - p.pad_x0, UP);
  const int ix_hi = floor_div((ox0 + TOW - 1) * DOWN + p.kw - 1 - p.pad_x0, UP);
  const int rows = iy_hi - iy_lo + 1, cols = ix_hi - ix_lo + 1;
  constexpr bool deint = DEINT;
  const int pitch = ufd_pitch(cols, deint);
  const int win = rows * pitch;
  const unsigned plane_in = (unsigned)(p.in_h * p.in_w);

  if (whole) {
    // Full-width tiles: the input of the workgroup is one contiguous run -- the PPB whole planes of a small map, or
    // (tiles_y > 1, PPB = 1) the band of input rows under TOH output rows of a large one.  Zero the windows (halo), then
    // scatter the run.
    const int nz = (PPB * win + 3) >> 2;
    for (int i = threadIdx.x; i < nz; i += 256) reinterpret_cast<float4*>(s_in)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nplanes = min(PPB, p.major - plane0);
    const int ys = tiles_y > 1 ? max(iy_lo, 0) : 0;                   // first input row of the run
    const int ye = tiles_y > 1 ? min(iy_hi, p.in_h - 1) : p.in_h - 1;
    const unsigned total = tiles_y > 1 ? (unsigned)(max(ye - ys + 1, 0) * p.in_w) : (unsigned)nplanes * plane_in;
    const float* src = p.in + (long)plane0 * plane_in + (long)ys * p.in_w;
    if (vec && total > 0) {             // (a band that lies wholly in the padding has total == 0: the plain branch stores the taps)
      for (unsigned base = 0; base < total; base += 256 * 4 * 4) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned e = base + 4 * (threadIdx.x + 256 * j);
          v[j] = *reinterpret_cast<const float4*>(src + (e < total ? e : 0));
        }
        if (base == 0) {
          if (threadIdx.x < p.kh * p.kw) s_k[tky * UFD_MAX_TAPS + tkx] = kv;
          __syncthreads();               // zero fill done before the scatter
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned e = base + 4 * (threadIdx.x + 256 * j);
          if (e >= total) continue;
          const unsigned pl = e / plane_in, rem = e - pl * plane_in;
          const int ry = (int)(rem / (unsigned)p.in_w), ix = (int)(rem - (unsigned)ry * p.in_w);
          const int r = ys + ry - iy_lo, c = ix - ix_lo;
          if (r < 0 || r >= rows) continue;
          float* d = s_in + pl * win + r * pitch;
          const float t[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (c + q >= 0 && c + q < cols) d[ufd_col(c + q, pitch, deint)] = t[q];
        }
      }
    } else {
      if (threadIdx.x < p.kh * p.kw) s_k[tky * UFD_MAX_TAPS + tkx] = kv;
      __syncthreads();
      for (unsigned e = threadIdx.x; e < total; e += 256) {
        const unsigned pl = e / plane_in, rem = e - pl * plane_in;
        const int ry = (int)(rem / (unsigned)p.in_w), ix = (int)(rem - (unsigned)ry * p.in_w);
        const int r = ys + ry - iy_lo, c = ix - ix_lo;
        if (r >= 0 && r < rows && c >= 0 && c < cols) s_in[pl * win + r * pitch + ufd_col(c, pitch, deint)] = src[e];
      }
    }
  } else {
    // One tile of a large plane (PPB == 1): coalesced row reads of the window with zero fill, 8 loads in flight.
    if (threadIdx.x < p.kh * p.kw) s_k[tky * UFD_MAX_TAPS + tkx] = kv;
    const float* src = p.in + (long)plane0 * plane_in;
    const unsigned total = (unsigned)(rows * cols);
    for (unsigned base = 0; base < total; base += 256 * 8) {
      float v[8]; int dst[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned e = base + threadIdx.x + 256 * j;
        const unsigned r = e / (unsigned)cols, c = e - r * (unsigned)cols;
        const int iy = iy_lo + (int)r, ix = ix_lo + (int)c;
        const bool ok = e < total && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        const float t = src[ok ? iy * p.in_w + ix : 0];
        v[j] = ok ? t : 0.f;
        dst[j] = e < total ? (int)(r * pitch) + ufd_col((int)c, pitch, deint) : -1;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (dst[j] >= 0) s_in[dst[j]] = v[j];
    }
  }
  __syncthreads();

  const long plane_out = (long)p.out_h * p.out_w;
  // Fast path for the 4 x 4 taps every live call uses (models/up_or_down_sampling.py:181-188): the taps live in
  // registers and a thread produces FOUR consecutive outputs of a row -- the first version computed one output per
  // thread and loop trip with the taps re-read from LDS and a floor division per tap (~40 instructions per output:
  // 1.6 TB/s on the 16x16 -> 32x32 up-sampling, instruction-bound at 20 % of HBM).  Per group of four outputs:
  // up 2: 8 LDS reads + 16 FMAs (every output sees 2 x 2 real samples), down 2: 40 + 64, plain FIR: 28 + 64; one
  // 16-byte store.
  if (k4) {
    float kr[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) kr[a][b] = s_k[a * UFD_MAX_TAPS + b];
    const int nq4 = nq >> 2;
    for (int q = 0; q < nq4; ++q) {
      const int o4 = threadIdx.x + 256 * q;
      const int tx = (o4 & ((TOW >> 2) - 1)) << 2;
      const int ty = (o4 >> (tow_log2 - 2)) & (TOH - 1);
      const int pl = o4 >> (tow_log2 - 2 + toh_log2);
      const int plane = plane0 + pl;
      const int oy = oy0 + ty, ox = ox0 + tx;
      const float* w_in = s_in + pl * win;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      if (UP == 2) {
        const int uy0 = oy - p.pad_y0, ux0 = ox - p.pad_x0;
        const int py = uy0 & 1, px = ux0 & 1;                       // first tap row / column that hits a real sample
        const float* r0 = w_in + (((uy0 + py) >> 1) - iy_lo) * pitch + (((ux0 + px) >> 1) - ix_lo);
        float v[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[a][c] = r0[a * pitch + c];
        // output j uses tap columns ((px + j) & 1) + 2 b at input columns ((j + ((px + j) & 1) - px) >> 1) + b
#define STK_UFD_UP(PY, PX)                                                                              \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                        \
    constexpr int dummy = 0; (void)dummy;                                                                \
    const int kx0 = ((PX) + j) & 1, c0 = (j + kx0 - (PX)) >> 1;                                          \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)            \
      acc[j] += v[a][c0 + b] * kr[(PY) + 2 * a][kx0 + 2 * b];                                            \
  }
        if (py == 0) { if (px == 0) { STK_UFD_UP(0, 0) } else { STK_UFD_UP(0, 1) } }
        else { if (px == 0) { STK_UFD_UP(1, 0) } else { STK_UFD_UP(1, 1) } }
#undef STK_UFD_UP
      } else if (DOWN == 2 && deint) {
        // first tap of output ox = window column 2 (ox - ox0): even, so its four outputs read even columns tx .. tx + 4 and
        // odd columns tx .. tx + 4 of the de-interleaved row (tx = ox - ox0, a multiple of 4: 16-byte reads)
        const float* r0 = w_in + (oy * 2 - p.pad_y0 - iy_lo) * pitch + tx;
        const int ph = pitch >> 1;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float4 e4 = *reinterpret_cast<const float4*>(r0 + a * pitch);
          const float4 o4v = *reinterpret_cast<const float4*>(r0 + a * pitch + ph);
          const float v[10] = {e4.x, o4v.x, e4.y, o4v.y, e4.z, o4v.z, e4.w, o4v.w, r0[a * pitch + 4], r0[a * pitch + ph + 4]};
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[j] += v[2 * j + b] * kr[a][b];
        }
      } else {
        constexpr int NC = DOWN == 2 ? 10 : 7;
        const float* r0 = w_in + (oy * DOWN - p.pad_y0 - iy_lo) * pitch + (ox * DOWN - p.pad_x0 - ix_lo);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float v[NC];
#pragma unroll
          for (int c = 0; c < NC; ++c) v[c] = r0[a * pitch + c];
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[j] += v[DOWN * j + b] * kr[a][b];
        }
      }
      if (plane < p.major && oy < p.out_h) {
        float* d = p.out + plane * plane_out + (long)oy * p.out_w + ox;
        if (ox + 3 < p.out_w && ((reinterpret_cast<uintptr_t>(d) & 15) == 0)) {
          float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
          if (p.beta != 0.f) {
            const float4 old = *reinterpret_cast<const float4*>(d);
            r.x += p.beta * old.x; r.y += p.beta * old.y; r.z += p.beta * old.z; r.w += p.beta * old.w;
          }
          *reinterpret_cast<float4*>(d) = r;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (ox + j < p.out_w) d[j] = (p.beta != 0.f ? p.beta * d[j] : 0.f) + acc[j];
        }
      }
    }
    return;
  }
#pragma unroll 4
  for (int q = 0; q < nq; ++q) {
    const int o = threadIdx.x + 256 * q;
    const int tx = o & (TOW - 1);
    const int ty = (o >> tow_log2) & (TOH - 1);
    const int pl = o >> (tow_log2 + toh_log2);
    const int plane = plane0 + pl;
    const int oy = oy0 + ty, ox = ox0 + tx;
    const float* w_in = s_in + pl * win;
    const int uy0 = oy * DOWN - p.pad_y0, ux0 = ox * DOWN - p.pad_x0;
    float acc = 0.f;
    // only taps whose upsampled coordinate is a real sample: ky = (-uy0) mod UP, stepping by UP
    const int ky0 = (UP - (uy0 & (UP - 1))) & (UP - 1), kx0 = (UP - (ux0 & (UP - 1))) & (UP - 1);
    for (int ky = ky0; ky < p.kh; ky += UP) {
      const float* row = w_in + (floor_div(uy0 + ky, UP) - iy_lo) * pitch;
      for (int kx = kx0; kx < p.kw; kx += UP)
        acc += row[ufd_col(floor_div(ux0 + kx, UP) - ix_lo, pitch, deint)] * s_k[ky * UFD_MAX_TAPS + kx];
    }
    if (plane < p.major && oy < p.out_h && ox < p.out_w) {
      float* d = p.out + plane * plane_out + (long)oy * p.out_w + ox;
      *d = (p.beta != 0.f ? p.beta * *d : 0.f) + acc;
    }
  }
}

// ---- plain 4 x 4 FIR whose rows are not a whole number of 16-byte pieces (round 4) ----------------------------------
// The pre-filter of conv_downsample_2d (models/up_or_down_sampling.py:144-178: pad (2, 2), no resampling) turns an H x H map
// into (H + 1) x (H + 1): rows of 65 / 33 / 17 floats.  In the tiled kernel that shape half-fills a power-of-two tile and its
// four-outputs-per-thread stores fall back to single floats 16 bytes apart (0.17-0.30 of HBM).  Here a workgroup owns PPB whole
// planes or a band of TOH rows (any multiple of four), staged like the tiled kernel's contiguous run, and walks its outputs in
// FLAT order along the rows: consecutive lanes = consecutive addresses of the output (the band is one contiguous piece of the
// plane), stores coalesce whatever the row length, LDS reads are stride-1; taps in registers.
__global__ __launch_bounds__(256) void upfirdn2d_fir_flat(UfdParams p, int toh, int ppb, int tiles_y, int vec) {
  extern __shared__ __attribute__((aligned(16))) float s_ufd[];
  float* s_k = s_ufd;
  float* s_in = s_ufd + UFD_MAX_TAPS * UFD_MAX_TAPS;
  const int ty_tile = blockIdx.x % tiles_y;
  const int plane0 = (blockIdx.x / tiles_y) * ppb;
  const int oy0 = ty_tile * toh;
  const int th = min(toh, p.out_h - oy0);                 // output rows of this band
  float kv = 0.f;
  const int tky = threadIdx.x >> 2, tkx = threadIdx.x & 3;
  if (threadIdx.x < 16) kv = p.k[(3 - tky) * 4 + (3 - tkx)];
  const int iy_lo = oy0 - p.pad_y0, ix_lo = -p.pad_x0;
  const int rows = ((th + 3) & ~3) + 3, cols = p.out_w + 3;      // whole quads of output rows (see below)
  const int pitch = cols | 1;
  const int win = rows * pitch;
  const unsigned plane_in = (unsigned)(p.in_h * p.in_w);
  const int nz = (ppb * win + 3) >> 2;
  for (int i = threadIdx.x; i < nz; i += 256) reinterpret_cast<float4*>(s_in)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nplanes = min(ppb, p.major - plane0);
  const int ys = tiles_y > 1 ? max(iy_lo, 0) : 0;
  const int ye = tiles_y > 1 ? min(iy_lo + rows - 1, p.in_h - 1) : p.in_h - 1;
  const unsigned total_in = tiles_y > 1 ? (unsigned)(max(ye - ys + 1, 0) * p.in_w) : (unsigned)nplanes * plane_in;
  const float* src = p.in + (long)plane0 * plane_in + (long)ys * p.in_w;
  if (threadIdx.x < 16) s_k[tky * UFD_MAX_TAPS + tkx] = kv;
  __syncthreads();                                        // zero fill done before the scatter
  if (vec) {
    for (unsigned base = 0; base < total_in; base += 256 * 4 * 4) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned e = base + 4 * (threadIdx.x + 256 * j);
        v[j] = *reinterpret_cast<const float4*>(src + (e < total_in ? e : 0));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned e = base + 4 * (threadIdx.x + 256 * j);
        if (e >= total_in) continue;
        const unsigned pl = e / plane_in, rem = e - pl * plane_in;
        const int ry = (int)(rem / (unsigned)p.in_w), ix = (int)(rem - (unsigned)ry * p.in_w);
        const int r = ys + ry - iy_lo, c = ix - ix_lo;
        if (r < 0 || r >= rows) continue;
        float* d = s_in + pl * win + r * pitch + c;
        const float t[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c + q >= 0 && c + q < cols) d[q] = t[q];
      }
    }
  } else {
    for (unsigned e = threadIdx.x; e < total_in; e += 256) {
      const unsigned pl = e / plane_in, rem = e - pl * plane_in;
      const int ry = (int)(rem / (unsigned)p.in_w), ix = (int)(rem - (unsigned)ry * p.in_w);
      const int r = ys + ry - iy_lo, c = ix - ix_lo;
      if (r >= 0 && r < rows && c >= 0 && c < cols) s_in[pl * win + r * pitch + c] = src[e];
    }
  }
  __syncthreads();
  float kr[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) kr[a][b] = s_k[a * UFD_MAX_TAPS + b];
  // a thread owns FOUR vertically adjacent outputs (rows 4 tq .. 4 tq + 3 of the band, one column): 7 x 4 window values serve
  // 64 FMAs (7 LDS reads per output instead of 16), and each of its four stores is lane-contiguous along the row
  const int nq = (th + 3) >> 2;                           // row quads of the band (the window has 4 nq + 3 rows)
  const int per_plane = nq * p.out_w;
  const int total = nplanes * per_plane;
  const float inv_pp = 1.f / (float)per_plane, inv_ow = 1.f / (float)p.out_w;
  const long plane_out = (long)p.out_h * p.out_w;
  for (int f = threadIdx.x; f < total; f += 256) {
    // f = (pl, tq, tx): quotients by float reciprocal, corrected by one step (exact: f < 2^21)
    int pl = (int)((float)f * inv_pp);
    int r = f - pl * per_plane;
    if (r < 0) { --pl; r += per_plane; } else if (r >= per_plane) { ++pl; r -= per_plane; }
    int tq = (int)((float)r * inv_ow);
    int tx = r - tq * p.out_w;
    if (tx < 0) { --tq; tx += p.out_w; } else if (tx >= p.out_w) { ++tq; tx -= p.out_w; }
    const float* w = s_in + pl * win + 4 * tq * pitch + tx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 7; ++a) {
      const float v0 = w[a * pitch], v1 = w[a * pitch + 1], v2 = w[a * pitch + 2], v3 = w[a * pitch + 3];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (a - j >= 0 && a - j < 4) {
          acc[j] += v0 * kr[a - j][0]; acc[j] += v1 * kr[a - j][1]; acc[j] += v2 * kr[a - j][2]; acc[j] += v3 * kr[a - j][3];
        }
    }
    float* d = p.out + (plane0 + pl) * plane_out + (long)(oy0 + 4 * tq) * p.out_w + tx;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * tq + j < th) d[(long)j * p.out_w] = (p.beta != 0.f ? p.beta * d[(long)j * p.out_w] : 0.f) + acc[j];
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
                        (long)in_h * in_w < 0x7fffffffL;
  // plain 4 x 4 FIR with rows that are not a whole number of 16-byte pieces: flat-order kernel
  if (tiled_ok && up_x == 1 && down_x == 1 && kh == 4 && kw == 4 && (p.out_w & 3) != 0 && p.out_w + 3 <= 1024 &&
      (long)p.out_h * p.out_w < (1L << 21)) {
    const int pitch = (p.out_w + 3) | 1;
    // ~2048 outputs per workgroup: whole planes while they fit the LDS budget, else bands of rows
    int toh = p.out_h, ppb = 1;
    const long plane_win = (long)(((p.out_h + 3) & ~3) + 3) * pitch;
    if (plane_win <= UFD_LDS_FLOATS) {
      ppb = (int)(2048 / ((long)p.out_h * p.out_w));
      if (ppb < 1) ppb = 1;
      while (ppb > 1 && (ppb * plane_win > UFD_LDS_FLOATS || ppb > 2 * major)) --ppb;
      if ((long)ppb * p.out_h * p.out_w >= (1L << 21)) ppb = 1;
    } else {
      toh = 2048 / p.out_w;
      if (toh < 1) toh = 1;
      toh &= ~3;                                           // bands of whole row quads
      if (toh < 4) toh = 4;
      while (toh > 4 && (long)(toh + 3) * pitch > UFD_LDS_FLOATS) toh -= 4;
    }
    const int tiles_y = stk_cdiv(p.out_h, toh);
    const long win = (long)(((toh + 3) & ~3) + 3) * pitch;
    const long nblk = (long)stk_cdiv(major, ppb) * tiles_y;
    if (win * ppb <= UFD_LDS_FLOATS && nblk <= 0x7fffffffL) {
      const int vec = (((long)in_h * in_w) & 3) == 0 && (in_w & 3) == 0 && stk_aligned16(input);
      const size_t shm = (UFD_MAX_TAPS * UFD_MAX_TAPS + (((size_t)win * ppb + 3) & ~(size_t)3)) * sizeof(float);
      hipLaunchKernelGGL(upfirdn2d_fir_flat, dim3((unsigned)nblk), dim3(256), shm, stream, p, toh, ppb, tiles_y, vec);
      STK_CHECK_LAUNCH();
      return STK_OK;
    }
  }
  if (tiled_ok) {
    // tile = TOH x TOW outputs (powers of two), PPB planes per workgroup.  Rows up to 256 outputs wide are taken whole
    // (tiles_x = 1): the workgroup's input is then one contiguous run of memory -- whole planes of a small map, a band of
    // rows of a large one -- staged with 16-byte loads (the 64-wide tiles of wider rows stage element by element: 24 % of
    // the HBM rate on the 64x64 -> 128x128 up-sampling against 74 % for the run form, profiles/r04_upfirdn2d.txt).
    // Outputs per workgroup: 1024 (4 per thread); 4096 for up-sampling, whose input is 4x smaller; 2048 for bands of
    // a down-sampling (the halo rows of a band are read twice: fewer, taller bands).
    const int tow_max = p.out_w <= 256 ? 8 : 6;
    int tow_log2 = 2, toh_log2 = 2;
    while (tow_log2 < tow_max && (1 << tow_log2) < p.out_w) ++tow_log2;
    const bool wide = (1 << tow_log2) >= p.out_w;                    // tiles_x == 1
    int out_log2 = 10;
    if (wide && up_x == 2) out_log2 = 12;
    else if (wide && down_x == 2 && p.out_h > (1 << (10 - tow_log2))) out_log2 = 11;
    while (toh_log2 + tow_log2 < out_log2 && (1 << toh_log2) < p.out_h) ++toh_log2;
    // LDS budget: PPB windows of rows x (cols | 1) floats
    const int up = up_x, down = down_x;
    auto win_floats = [&](int th_log2) {
      const long r = (((1 << th_log2) - 1) * down + kh - 1) / up + 2, c = (((1 << tow_log2) - 1) * down + kw - 1) / up + 2;
      return r * ufd_pitch((int)c, ufd_deint(down, tow_log2));
    };
    while (toh_log2 > 2 && win_floats(toh_log2) > UFD_LDS_FLOATS) --toh_log2;
    const int TOW = 1 << tow_log2, TOH = 1 << toh_log2;
    const int tiles_x = stk_cdiv(p.out_w, TOW), tiles_y = stk_cdiv(p.out_h, TOH);
    const int whole = tiles_x == 1;                                  // contiguous-run staging (kernel: `whole`)
    int ppb_log2 = tiles_x * tiles_y == 1 ? max(out_log2 - tow_log2 - toh_log2, 0) : 0;
    const int rows = ((TOH - 1) * down + kh - 1) / up + 2, cols = ((TOW - 1) * down + kw - 1) / up + 2;
    const long win = (long)rows * ufd_pitch(cols, ufd_deint(down, tow_log2));
    while (ppb_log2 > 0 && ((win << ppb_log2) > UFD_LDS_FLOATS || (1 << ppb_log2) > 2 * major))
      --ppb_log2;
    const int nq = (1 << (ppb_log2 + tow_log2 + toh_log2)) >> 8;
    const long groups = ((long)major + (1 << ppb_log2) - 1) >> ppb_log2;
    const long nblk = groups * tiles_x * tiles_y;
    if (win <= UFD_LDS_FLOATS && nblk <= 0x7fffffffL && nq >= 1) {
      const int vec = whole && (((long)in_h * in_w) & 3) == 0 && (in_w & 3) == 0 && stk_aligned16(input);
      const size_t shm = (UFD_MAX_TAPS * UFD_MAX_TAPS + (((size_t)win << ppb_log2) + 3 & ~(size_t)3)) * sizeof(float);
      // four outputs per thread and trip (taps in registers) for the 4 x 4 FIR; windows are sized so that the reads of a
      // group that hangs over the tile edge stay inside the staged window (rows / cols above have one spare)
      const int k4 = kh == 4 && kw == 4 && (nq & 3) == 0 && tow_log2 >= 2;
      dim3 grid((unsigned)nblk), block(256);
#define STK_UFD(U, D, I)                                                                                               \
  hipLaunchKernelGGL((upfirdn2d_tiled<U, D, I>), grid, block, shm, stream, p, tow_log2, toh_log2, ppb_log2, nq, tiles_x, \
                     tiles_y, whole, vec, k4)
      if (up_x == 1 && down_x == 1) STK_UFD(1, 1, false);
      else if (up_x == 1 && ufd_deint(down_x, tow_log2)) STK_UFD(1, 2, true);
      else if (up_x == 1) STK_UFD(1, 2, false);
      else STK_UFD(2, 1, false);
#undef STK_UFD
      STK_CHECK_LAUNCH();
      return STK_OK;
    }
  }
  const long total = (long)major * p.out_h * p.out_w * minor;
  hipLaunchKernelGGL(upfirdn2d_direct, dim3(stk_ew_grid(total)), dim3(256), 0, stream, p);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

// ---- other floating types (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, op/upfirdn2d_kernel.cu:311):
// one thread per output element, any parameters; half accumulates in fp32 and rounds once, double in double.  Nothing
// on the hot path feeds these -- they complete the drop-in surface of the native op.
template <class T, class ACC>
__global__ __launch_bounds__(256) void upfirdn2d_direct_t(const T* __restrict__ in, const T* __restrict__ k, T* __restrict__ out,
                                                          UfdParams p) {
  const long total = (long)p.major * p.out_h * p.out_w * p.minor;
  const long stride = (long)gridDim.x * 256;
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += stride) {
    long r = o;
    const int mi = (int)(r % p.minor); r /= p.minor;
    const int ox = (int)(r % p.out_w); r /= p.out_w;
    const int oy = (int)(r % p.out_h);
    const int mj = (int)(r / p.out_h);
    ACC acc = 0;
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
        acc += (ACC)in[((long)(mj * p.in_h + iy) * p.in_w + ix) * p.minor + mi] * (ACC)k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
      }
    }
    out[o] = (T)acc;
  }
}

template <class T, class ACC>
int launch_t(const void* input, const void* kernel, void* out, int major, int in_h, int in_w, int minor, int kh, int kw,
             int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, hipStream_t stream) {
  if (!input || !kernel || !out || major <= 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0 ||
      up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0)
    return STK_EINVAL;
  UfdParams p = {};
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  if (p.out_h <= 0 || p.out_w <= 0) return STK_EINVAL;
  const long total = (long)major * p.out_h * p.out_w * minor;
  hipLaunchKernelGGL((upfirdn2d_direct_t<T, ACC>), dim3(stk_ew_grid(total)), dim3(256), 0, stream, static_cast<const T*>(input),
                     static_cast<const T*>(kernel), static_cast<T*>(out), p);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

}  // namespace

extern "C" {

int stk_upfirdn2d_f16(const void* input, const void* kernel, void* out, int major, int in_h, int in_w, int minor, int kh,
                      int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                      void* stream) {
  return launch_t<_Float16, float>(input, kernel, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                   pad_y0, pad_y1, (hipStream_t)stream);
}

int stk_upfirdn2d_f64(const double* input, const double* kernel, double* out, int major, int in_h, int in_w, int minor, int kh,
                      int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                      void* stream) {
  return launch_t<double, double>(input, kernel, out, major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                  pad_y0, pad_y1, (hipStream_t)stream);
}

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
