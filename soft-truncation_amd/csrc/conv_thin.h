// conv_thin.h -- convolutions with a "thin" side (<= 4 channels), for gfx950.  Included by conv.hip inside its
// anonymous namespace.
//
// The score networks touch 3-channel tensors at full resolution: the stem (3 -> nf), the head (nf -> 3), and in the
// NCSN++ pyramids the `output_skip` 3x3 convs (C -> 3) and the `input_skip` 1x1 Combine convs (3 -> C)
// (models/ncsnpp.py:219-252, 378-422).  As a GEMM one dimension is 3: a 64-row MFMA tile wastes 21/22 of its work and
// the generic kernel ran those layers at 2-16 TFLOP/s (217 us for the CIFAR stem, ~100 us each for the 256x256
// pyramid layers) although they only stream their big tensor once.  Here they are what they are -- streaming kernels:
//
//   thin_in_kernel    out[o, p] = sum_{c < CT, tap} W(o, c, tap) * thin[c, p + tap]       CT <= 4 input channels
//                     one thread per pixel holds its CT x taps patch in registers, loops over the output channels
//                     with the weights broadcast from LDS, and writes each output plane coalesced.
//                     Forward of the stem / Combine layers, and the DATA GRADIENT of the thin-output layers
//                     (there the thin tensor is dy and the weights are read transposed and tap-flipped).
//   thin_out_kernel   out[o < CO, p] = sum_{ci, tap} w[o, ci, tap] * x[ci, p + tap]        CO <= 4 output channels
//                     one thread per pixel, CO accumulators, weights broadcast from LDS in blocks of 128 channels;
//                     the nine shifted reads of a channel plane overlap between neighbouring lanes and hit L1.
//
//   thin_wgrad_kernel  the weight gradient of either kind (below).
//
// Both are HBM-bound on the big tensor (4 bytes per element, once).  Epilogues are the same arithmetic as EpFwd /
// EpDgrad.  3x3 / stride 1 / pad 1 and 1x1 / stride 1 / pad 0 only, [Cout,Cin,kh,kw] weights.
#pragma once

namespace thin {

constexpr int MAXK = 36;          // CT * taps <= 4 * 9
constexpr int OB = 32;            // output channels per workgroup of thin_in_kernel: N*HW/256 * OUT/32 workgroups keep
                                  // >= 8 waves per SIMD in flight (with 128 the kernel ran 2 waves per SIMD, latency-bound)

struct Args {
  const float* thin;   // the <= 4 channel tensor [N, CT, H, W]
  const float* w;
  long so, sc;         // weight(o, c, tap) = w[o*so + c*sc + (flip ? taps-1-tap : tap)]
  int flip, CT, OUT, taps;
  int dgrad;           // 0: forward epilogue (bias, temb, res, div -> y);  1: dgrad epilogue (beta/alpha -> dx1/dx2)
};

// halo mask and offsets of the taps of pixel (y, x): bit t set <=> tap t reads inside the image
__device__ __forceinline__ unsigned tap_mask(int y, int x, int H, int W, int taps) {
  if (taps == 1) return 1u;
  unsigned m = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) m |= 1u << t;
  }
  return m;
}

template <int TAPS>
__global__ __launch_bounds__(256) void thin_in_kernel(ConvP p, Args a) {
  __shared__ __attribute__((aligned(16))) float w_s[OB * MAXK];
  const int K = a.CT * TAPS;
  const int o0 = blockIdx.y * OB, no = min(OB, a.OUT - o0);
  // weights of this block's output channels: w_s[o][c*TAPS + t]
  for (int i = threadIdx.x; i < no * 4 * TAPS; i += 256) {
    const int o = i / (4 * TAPS), k = i - o * 4 * TAPS, c = k / TAPS, t = k - c * TAPS;
    w_s[o * MAXK + k] = k < K ? a.w[(long)(o0 + o) * a.so + c * a.sc + (a.flip ? TAPS - 1 - t : t)] : 0.f;
  }
  const long n = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = n < (long)p.N * p.HW;
  const int b = live ? (int)(n / p.HW) : 0, hw = live ? (int)(n - (long)b * p.HW) : 0;
  const int y = hw / p.W, x = hw - y * p.W;
  const unsigned mask = live ? tap_mask(y, x, p.H, p.W, TAPS) : 0u;
  float xv[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) xv[k] = 0.f;
  {
    const float* base = a.thin + (long)b * a.CT * p.HW + hw;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const bool ok = c < a.CT && ((mask >> t) & 1u);
        const int off = TAPS == 9 ? (t / 3 - 1) * p.W + (t % 3 - 1) : 0;
        const float v = base[ok ? (long)c * p.HW + off : 0];       // unconditional load from a safe address
        xv[c * TAPS + t] = ok ? v : 0.f;
      }
  }
  __syncthreads();
  if (!live) return;
  const float inv_div = p.inv_div;
  for (int o = 0; o < no; ++o) {
    // the weight row is the same for every lane: 16-byte broadcast reads (a ds_read_b32 per weight made this
    // kernel LDS-issue-bound).  k >= K: both factors are 0.
    const float4* wr = reinterpret_cast<const float4*>(w_s + o * MAXK);
    float acc = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < (4 * TAPS + 3) / 4; ++k4) {
      const float4 wv = wr[k4];
      acc += wv.x * xv[4 * k4] + wv.y * xv[4 * k4 + 1] + wv.z * xv[4 * k4 + 2] + wv.w * xv[4 * k4 + 3];
    }
    const int oc = o0 + o;
    if (!a.dgrad) {
      const long idx = ((long)b * a.OUT + oc) * p.HW + hw;
      float v = acc;
      if (p.bias) v += p.bias[oc];
      if (p.temb) v += p.temb[(long)b * p.temb_stride + oc];
      if (p.res) v += p.res[idx];
      if (p.use_div) v *= inv_div;
      p.y[idx] = v;
    } else {
      float* d; float beta;
      if (oc < p.C1) { d = p.dx1 ? p.dx1 + ((long)b * p.C1 + oc) * p.HW + hw : nullptr; beta = p.beta1; }
      else { d = p.dx2 ? p.dx2 + ((long)b * p.C2 + (oc - p.C1)) * p.HW + hw : nullptr; beta = p.beta2; }
      if (d) *d = (beta != 0.f ? beta * *d : 0.f) + p.alpha * acc;
    }
  }
}

// Output channels CO <= 4; input = concat(x1, x2).  A workgroup is 64 pixels x 4 waves; wave q reduces channels
// [32q, 32q+32) of every 128-channel block (four times the waves in flight of a thread-per-pixel layout, which was
// latency-bound on the plane reads), and the four partial sums meet in LDS in a fixed order.
template <int TAPS>
__global__ __launch_bounds__(256) void thin_out_kernel(ConvP p, int CO) {
  __shared__ __attribute__((aligned(16))) float w_s[128 * 9 * 4];     // [c][tap][o]: the 4 output weights of a tap in one read
  __shared__ float red[4][4][64];                                     // [wave][o][pixel]
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long n = (long)blockIdx.x * 64 + lane;
  const bool live = n < (long)p.N * p.HW;
  const int b = live ? (int)(n / p.HW) : 0, hw = live ? (int)(n - (long)b * p.HW) : 0;
  const int y = hw / p.W, x = hw - y * p.W;
  const unsigned mask = live ? tap_mask(y, x, p.H, p.W, TAPS) : 0u;
  int off[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) off[t] = ((mask >> t) & 1u) ? (TAPS == 9 ? (t / 3 - 1) * p.W + (t % 3 - 1) : 0) : 0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < p.Cin; c0 += 128) {
    const int nc = min(128, p.Cin - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * nc * TAPS; i += 256) {
      const int o = i & 3, r = i >> 2;                                // r = c*TAPS + t
      w_s[r * 4 + o] = o < CO ? p.w[((long)o * p.Cin + c0) * TAPS + r] : 0.f;
    }
    __syncthreads();
    const int ce = min(nc, q * 32 + 32);
#pragma unroll 2
    for (int c = q * 32; c < ce; ++c) {
      const int ci = c0 + c;
      const float* plane = ci < p.C1 ? p.x1 + ((long)b * p.C1 + ci) * p.HW + hw
                                     : p.x2 + ((long)b * p.C2 + (ci - p.C1)) * p.HW + hw;
      float v[TAPS];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) v[t] = plane[off[t]];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const float xv = ((mask >> t) & 1u) ? v[t] : 0.f;
        const float4 wv = *reinterpret_cast<const float4*>(w_s + (c * TAPS + t) * 4);     // o >= CO: zero weights
        acc[0] += wv.x * xv; acc[1] += wv.y * xv; acc[2] += wv.z * xv; acc[3] += wv.w * xv;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) red[q][o][lane] = acc[o];
  __syncthreads();
  if (!live || q >= CO) return;
  const int o = q;                                                    // wave q finishes output channel q
  float v = ((red[0][o][lane] + red[1][o][lane]) + red[2][o][lane]) + red[3][o][lane];
  const long idx = ((long)b * CO + o) * p.HW + hw;
  if (p.bias) v += p.bias[o];
  if (p.temb) v += p.temb[(long)b * p.temb_stride + o];
  if (p.res) v += p.res[idx];
  if (p.use_div) v *= p.inv_div;
  p.y[idx] = v;
}

// Weight gradient of a layer with a thin side:  dw[big channel, thin channel, tap] = sum over (n, pixel) of
// big[n, bc, q] * thin[n, tc, q + off(tap)].  For the stem / Combine layers big = dy and thin = x (tap as is); for the
// thin-OUTPUT layers big = x (a concat of two tensors at most), thin = dy and q + off(t) pairs with the weight tap
// taps-1-t.  A workgroup owns four big channels and walks 256-pixel items of the flattened (image, pixel block) list
// with stride gridDim.x; a thread keeps 4 x (CT*taps) sums in registers, fed by one patch of the thin tensor (L1/L2
// hits) and four coalesced reads of the big tensor -- which is read exactly once overall.  The 256 per-thread sums
// of a workgroup meet in LDS (per wave, conflict-free pitch 65, then across the four waves in a fixed order) and go
// to slab blockIdx.x in the [tap][co][ci] layout of splitk_reduce_kernel.
struct WArgs {
  const float* thin; const float* b1; const float* b2;
  int B1, B2;          // channels of the big tensor's two sources (B2 = 0: one source)
  int CT;              // thin channels
  int head;            // 0: thin = x (dw[bc][tc][tap]);  1: thin = dy (dw[tc][bc][taps-1-tap])
  int items, pxb;      // items = N * pxb, pxb = ceil(HW / 256)
  float* part; long slab;
};

template <int TAPS, int CTT>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(ConvP p, WArgs a) {
  constexpr int KK = CTT * TAPS;
  __shared__ float red[4][KK * 65];
  __shared__ float wsum[4][4 * KK];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int BC = a.B1 + a.B2, bc0 = blockIdx.y * 4;
  float acc[4][KK];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < KK; ++k) acc[j][k] = 0.f;
  for (int it = blockIdx.x; it < a.items; it += gridDim.x) {
    const int n = it / a.pxb, hw = (it - n * a.pxb) * 256 + tid;
    const bool live = hw < p.HW;
    const int y = hw / p.W, x = hw - y * p.W;
    const unsigned mask = live ? tap_mask(y, x, p.H, p.W, TAPS) : 0u;
    const float* tb = a.thin + (long)n * a.CT * p.HW + (live ? hw : 0);
    float pv[KK];
#pragma unroll
    for (int c = 0; c < CTT; ++c)
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const bool ok = c < a.CT && ((mask >> t) & 1u);
        const int off = TAPS == 9 ? (t / 3 - 1) * p.W + (t % 3 - 1) : 0;
        const float v = tb[ok ? (long)c * p.HW + off : 0];
        pv[c * TAPS + t] = ok ? v : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bc = bc0 + j;
      if (bc < BC) {                                                  // uniform
        const float* bp = bc < a.B1 ? a.b1 + ((long)n * a.B1 + bc) * p.HW : a.b2 + ((long)n * a.B2 + (bc - a.B1)) * p.HW;
        const float v = live ? bp[hw] : 0.f;
#pragma unroll
        for (int k = 0; k < KK; ++k) acc[j][k] += v * pv[k];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int k = 0; k < KK; ++k) red[wv][k * 65 + lane] = acc[j][k];
    __syncthreads();
    if (lane < KK) {
      float sum = 0.f;
      for (int i = 0; i < 64; ++i) sum += red[wv][lane * 65 + i];
      wsum[wv][j * KK + lane] = sum;
    }
    __syncthreads();
  }
  if (tid < 4 * KK) {
    const int j = tid / KK, k = tid - j * KK, c = k / TAPS, t = k - c * TAPS, bc = bc0 + j;
    if (bc < BC && c < a.CT) {
      const float total = ((wsum[0][tid] + wsum[1][tid]) + wsum[2][tid]) + wsum[3][tid];
      const long idx = a.head ? ((long)(TAPS - 1 - t) * a.CT + c) * BC + bc : ((long)t * BC + bc) * a.CT + c;
      a.part[(long)blockIdx.x * a.slab + idx] = total;
    }
  }
}

// number of slabs (= gridDim.x) of thin_wgrad_kernel: ~1024 workgroups over the chip, never more than the work items
inline int wgrad_slabs(int N, long HW, int BC) {
  const long items = (long)N * ((HW + 255) / 256);
  long s = 1024 / ((BC + 3) / 4);
  if (s < 1) s = 1;
  return (int)(s < items ? s : items);
}

inline bool geometry_ok(const ConvP& p) {
  return p.stride == 1 && p.OH == p.H && p.OW == p.W && ((p.taps == 9 && p.pad == 1) || (p.taps == 1 && p.pad == 0)) &&
         p.w_layout == 0;
}

}  // namespace thin
