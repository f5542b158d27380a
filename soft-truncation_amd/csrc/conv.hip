// conv.hip -- convolution forward / data-gradient / weight-gradient as implicit GEMM on the fp32
// MFMA core (igemm.h), plus the batched strided GEMM entry point, for gfx950.
//
// Replaces every nn.Conv2d / NIN / nn.Linear / einsum call of the score network
// (reference models/layerspp.py:95-99,273-287; models/layers.py:100-124,546-555).
//
//   forward   M = Cout        N = batch*OH*OW   K = Cin*KH*KW    A = weights        B = im2col(x)
//   dgrad     M = Cin         N = batch*H*W     K = Cout*KH*KW   A = weights^T      B = gather(dy)
//   wgrad     M = Cout        N = Cin           K = batch*OH*OW  A = dy             B = shifted x   (one GEMM per tap)
//
// In all three the pixel index is the fastest-varying index of the activation operand in HBM (NCHW),
// so the activation loaders put lanes along pixels: a wave reads 64 consecutive floats (256 B) per
// instruction, shifted by the tap offset; halo / padding / stride-2 holes are predicated zeros.
// The channel-concat of the up path (two sources) is just a pointer select on the channel index.
// Weights are read in their native layouts ([Cout,Cin,KH,KW] or NIN's [Cin,Cout]) by whichever lane
// mapping makes those reads contiguous; they are L2-resident (<= 4.7 MB per layer).
//
// wgrad reduces over batch*OH*OW (131072 for a 32x32 map at batch 128) into a small output, so it is
// split-K: each slice writes its partial tile to a scratch slab and a second kernel sums the slabs in a
// fixed order into dw (deterministic -- no float atomics).
#include "igemm.h"

namespace {

using igemm::Cfg;
using igemm::KMajor;
using igemm::MnMajor;

struct ConvP {
  const float* x1; const float* x2; int C1, C2;
  const float* w; int w_layout;
  const float* bias; const float* temb; int temb_stride; const float* res; float inv_div; int use_div;
  float* y;
  const float* dy; float* dx1; float* dx2; float beta1, beta2, alpha;
  float* part; long part_stride;
  int N, H, W, Cin, Cout, OH, OW, KH, KW, stride, pad, sshift;
  int HW, OHW, taps;
  // generic strided A operand (weights viewed as a matrix): element (m,k) at wA[m*sam + k*sak]
  long sam, sak;
};

// ---- A loaders ------------------------------------------------------------------------------------------
// Strided matrix A(m,k) = w[m*sam + k*sak], M rows, Ktot columns.
template <class C, bool K_CONTIG>
struct AStrided {
  int m0, tid, M, Ktot;
  __device__ void init(const ConvP& p, int m0_, int tid_, int) {
    m0 = m0_; tid = tid_;
    M = 0; Ktot = 0;
  }
  __device__ __forceinline__ void set_dims(int M_, int K_) { M = M_; Ktot = K_; }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NA]) {
    if (K_CONTIG) {
      using Mp = KMajor<C::BM, C::KC>;
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        const int m = m0 + Mp::mn(tid, i), k = k0 + Mp::kk(tid, i);
        r[i] = (m < M && k < Ktot) ? p.w[(long)m * p.sam + (long)k * p.sak] : 0.f;
      }
    } else {
      using Mp = MnMajor<C::BM, C::KC>;
      const int m = m0 + Mp::mn(tid);
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        const int k = k0 + Mp::kk(tid, i);
        r[i] = (m < M && k < Ktot) ? p.w[(long)m * p.sam + (long)k * p.sak] : 0.f;
      }
    }
  }
  __device__ void store(const float (&r)[C::NA], float* t) {
    if (K_CONTIG) igemm::store_k_major<C::BM, C::KC, C::LDA>(r, t, tid);
    else igemm::store_mn_major<C::BM, C::KC, C::LDA>(r, t, tid);
  }
};
template <class C, bool K_CONTIG>
struct AFwd : AStrided<C, K_CONTIG> {      // M = Cout, K = Cin*taps
  __device__ void init(const ConvP& p, int m0_, int tid_, int zb) {
    AStrided<C, K_CONTIG>::init(p, m0_, tid_, zb);
    this->set_dims(p.Cout, p.Cin * p.taps);
  }
};
template <class C, bool K_CONTIG>
struct ADgrad1 : AStrided<C, K_CONTIG> {   // 1x1 dgrad: M = Cin, K = Cout
  __device__ void init(const ConvP& p, int m0_, int tid_, int zb) {
    AStrided<C, K_CONTIG>::init(p, m0_, tid_, zb);
    this->set_dims(p.Cin, p.Cout);
  }
};
// 3x3 dgrad on [Cout,Cin,3,3]:  A(m=ci, k=(co,tap)) = w[(co*Cin + ci)*9 + tap]
template <class C>
struct ADgrad9 {
  int m0, tid;
  __device__ void init(const ConvP&, int m0_, int tid_, int) { m0 = m0_; tid = tid_; }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NA]) {
    using Mp = KMajor<C::BM, C::KC>;
    const int co0 = k0 / 9;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) {
      const int m = m0 + Mp::mn(tid, i), kk = Mp::kk(tid, i);
      const int co = co0 + kk / 9, tap = kk % 9;
      r[i] = (m < p.Cin && co < p.Cout) ? p.w[((long)co * p.Cin + m) * 9 + tap] : 0.f;
    }
  }
  __device__ void store(const float (&r)[C::NA], float* t) { igemm::store_k_major<C::BM, C::KC, C::LDA>(r, t, tid); }
};

// ---- B loaders: activations, lanes along pixels -----------------------------------------------------------
// forward im2col:  B(k=(ci,tap), n=(b,oy,ox)) = x[b, ci, oy*s + kh - pad, ox*s + kw - pad]
template <class C, int TAPS>
struct BFwd {
  using Mp = MnMajor<C::BN, C::KC>;
  static_assert(Mp::PER % TAPS == 0, "k rows per thread must cover whole channels");
  const float* p1; const float* p2;
  int cig, tid; unsigned mask;
  __device__ void init(const ConvP& p, int n0, int tid_, int) {
    tid = tid_;
    const int n = n0 + Mp::mn(tid);
    cig = Mp::kgroup(tid) * (Mp::PER / TAPS);
    mask = 0; p1 = p.x1; p2 = p.x2;
    if (n < p.N * p.OHW) {
      const int b = n / p.OHW, ohw = n - b * p.OHW;
      const int oy = ohw / p.OW, ox = ohw - oy * p.OW;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      const int KW = TAPS == 9 ? 3 : 1;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int iy = iy0 + t / KW, ix = ix0 + t % KW;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << t;
      }
      const long pix = (long)iy0 * p.W + ix0;
      p1 = p.x1 + (long)b * p.C1 * p.HW + pix;
      p2 = p.x2 ? p.x2 + (long)b * p.C2 * p.HW + pix : nullptr;
    }
  }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NB]) {
    const int ci0 = k0 / TAPS + cig;
    const int KW = TAPS == 9 ? 3 : 1;
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
      const int ci = ci0 + i / TAPS, t = i % TAPS;
      const int off = (t / KW) * p.W + (t % KW);
      float v = 0.f;
      if (((mask >> t) & 1u) && ci < p.Cin)
        v = ci < p.C1 ? p1[(long)ci * p.HW + off] : p2[(long)(ci - p.C1) * p.HW + off];
      r[i] = v;
    }
  }
  __device__ void store(const float (&r)[C::NB], float* t) { igemm::store_mn_major<C::BN, C::KC, C::LDB>(r, t, tid); }
};

// dgrad gather:  B(k=(co,tap), n=(b,y,x)) = dy[b, co, (y+pad-kh)/s, (x+pad-kw)/s]  where divisible & in range
template <class C, int TAPS>
struct BDgrad {
  using Mp = MnMajor<C::BN, C::KC>;
  static_assert(Mp::PER % TAPS == 0, "k rows per thread must cover whole channels");
  const float* pd;
  int cig, tid, yp, xp; unsigned mask;
  __device__ void init(const ConvP& p, int n0, int tid_, int) {
    tid = tid_;
    const int n = n0 + Mp::mn(tid);
    cig = Mp::kgroup(tid) * (Mp::PER / TAPS);
    mask = 0; pd = p.dy; yp = 0; xp = 0;
    if (n < p.N * p.HW) {
      const int b = n / p.HW, hw = n - b * p.HW;
      const int y = hw / p.W, x = hw - y * p.W;
      yp = y + p.pad; xp = x + p.pad;
      const int KW = TAPS == 9 ? 3 : 1;
      const int sm = p.stride - 1;     // stride is 1 or 2
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int ty = yp - t / KW, tx = xp - t % KW;
        if (ty >= 0 && tx >= 0 && !(ty & sm) && !(tx & sm) && (ty >> p.sshift) < p.OH && (tx >> p.sshift) < p.OW)
          mask |= 1u << t;
      }
      pd = p.dy + (long)b * p.Cout * p.OHW;
    }
  }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NB]) {
    const int co0 = k0 / TAPS + cig;
    const int KW = TAPS == 9 ? 3 : 1;
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
      const int co = co0 + i / TAPS, t = i % TAPS;
      float v = 0.f;
      if (((mask >> t) & 1u) && co < p.Cout) {
        const int off = ((yp - t / KW) >> p.sshift) * p.OW + ((xp - t % KW) >> p.sshift);
        v = pd[(long)co * p.OHW + off];
      }
      r[i] = v;
    }
  }
  __device__ void store(const float (&r)[C::NB], float* t) { igemm::store_mn_major<C::BN, C::KC, C::LDB>(r, t, tid); }
};

// ---- wgrad loaders (K = pixels, lanes along pixels; tap = blockIdx.z) ------------------------------------
// KMajor with KC = 32: kk = tid & 31 is fixed per thread, mn = tid/32 + 8 r.
template <class C>
struct AWgrad {       // A(m=co, k=pixel) = dy[b, co, ohw]
  static_assert(C::KC == 32, "wgrad uses 32-pixel chunks");
  int m0, tid;
  __device__ void init(const ConvP&, int m0_, int tid_, int) { m0 = m0_; tid = tid_; }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NA]) {
    const int k = k0 + (tid & 31);
    const bool kv = k < p.N * p.OHW;
    const int b = kv ? k / p.OHW : 0;
    const int ohw = k - b * p.OHW;
    const float* base = p.dy + (long)b * p.Cout * p.OHW + ohw;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) {
      const int m = m0 + (tid >> 5) + 8 * i;
      r[i] = (kv && m < p.Cout) ? base[(long)m * p.OHW] : 0.f;
    }
  }
  __device__ void store(const float (&r)[C::NA], float* t) {
#pragma unroll
    for (int i = 0; i < C::NA; ++i) t[(tid & 31) * C::LDA + (tid >> 5) + 8 * i] = r[i];
  }
};
template <class C>
struct BWgrad {       // B(k=pixel, n=ci) = x[b, ci, oy*s + kh - pad, ox*s + kw - pad]
  static_assert(C::KC == 32, "wgrad uses 32-pixel chunks");
  int n0, tid, kh, kw;
  __device__ void init(const ConvP& p, int n0_, int tid_, int zb) {
    n0 = n0_; tid = tid_;
    kh = zb / p.KW; kw = zb - kh * p.KW;
  }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NB]) {
    const int k = k0 + (tid & 31);
    bool kv = k < p.N * p.OHW;
    const int b = kv ? k / p.OHW : 0;
    const int ohw = k - b * p.OHW;
    const int oy = ohw / p.OW, ox = ohw - oy * p.OW;
    const int iy = oy * p.stride + kh - p.pad, ix = ox * p.stride + kw - p.pad;
    kv = kv && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const long pix = (long)iy * p.W + ix;
    const float* b1 = p.x1 + (long)b * p.C1 * p.HW + pix;
    const float* b2 = p.x2 ? p.x2 + (long)b * p.C2 * p.HW + pix : nullptr;
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
      const int ci = n0 + (tid >> 5) + 8 * i;
      float v = 0.f;
      if (kv && ci < p.Cin) v = ci < p.C1 ? b1[(long)ci * p.HW] : b2[(long)(ci - p.C1) * p.HW];
      r[i] = v;
    }
  }
  __device__ void store(const float (&r)[C::NB], float* t) {
#pragma unroll
    for (int i = 0; i < C::NB; ++i) t[(tid & 31) * C::LDB + (tid >> 5) + 8 * i] = r[i];
  }
};

// ---- epilogues ----------------------------------------------------------------------------------------------
struct EpFwd {        // y = (acc + bias + temb + res) * inv_div
  int b; long col_off;
  __device__ void init(const ConvP&, int, int) {}
  __device__ void col(const ConvP& p, int n) {
    b = n / p.OHW;
    col_off = (long)b * p.Cout * p.OHW + (n - b * p.OHW);
  }
  __device__ void put(const ConvP& p, int m, int, float acc) {
    const long idx = col_off + (long)m * p.OHW;
    float v = acc;
    if (p.bias) v += p.bias[m];
    if (p.temb) v += p.temb[(long)b * p.temb_stride + m];
    if (p.res) v += p.res[idx];
    if (p.use_div) v *= p.inv_div;
    p.y[idx] = v;
  }
};
struct EpDgrad {      // dx{1,2} = beta*dx + alpha*acc, rows routed to the two sources of the concat
  int b, hw;
  __device__ void init(const ConvP&, int, int) {}
  __device__ void col(const ConvP& p, int n) { b = n / p.HW; hw = n - b * p.HW; }
  __device__ void put(const ConvP& p, int m, int, float acc) {
    float* d; float beta;
    if (m < p.C1) { d = p.dx1 ? p.dx1 + ((long)b * p.C1 + m) * p.HW + hw : nullptr; beta = p.beta1; }
    else { d = p.dx2 ? p.dx2 + ((long)b * p.C2 + (m - p.C1)) * p.HW + hw : nullptr; beta = p.beta2; }
    if (d) *d = (beta != 0.f ? beta * *d : 0.f) + p.alpha * acc;
  }
};
struct EpWgrad {      // partial slab of split zs, tap zb, in the weight's own layout
  float* slab; int tap;
  __device__ void init(const ConvP& p, int zb, int zs) { slab = p.part + (long)zs * p.part_stride; tap = zb; }
  __device__ void col(const ConvP&, int) {}
  __device__ void put(const ConvP& p, int m, int n, float acc) {
    if (p.w_layout == 0) slab[((long)m * p.Cin + n) * p.taps + tap] = acc;
    else slab[(long)n * p.Cout + m] = acc;
  }
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                            long n, int splits, long stride, float alpha) {
  const long gstride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += gstride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += part[(long)z * stride + i];
    dw[i] += alpha * s;
  }
}

// ---- generic batched strided GEMM --------------------------------------------------------------------------
struct GemmP {
  const float* A; long sam, sak, sab;
  const float* B; long sbk, sbn, sbb;
  float* C; long scm, scn, scb;
  const float* bias; int bias_mode;
  int M, N, K;
  float alpha, beta;
};
template <class C, bool K_CONTIG>
struct GA {
  const float* base; int m0, tid;
  __device__ void init(const GemmP& p, int m0_, int tid_, int zb) { m0 = m0_; tid = tid_; base = p.A + (long)zb * p.sab; }
  __device__ void load(const GemmP& p, int k0, float (&r)[C::NA]) {
    if (K_CONTIG) {
      using Mp = KMajor<C::BM, C::KC>;
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        const int m = m0 + Mp::mn(tid, i), k = k0 + Mp::kk(tid, i);
        r[i] = (m < p.M && k < p.K) ? base[(long)m * p.sam + (long)k * p.sak] : 0.f;
      }
    } else {
      using Mp = MnMajor<C::BM, C::KC>;
      const int m = m0 + Mp::mn(tid);
#pragma unroll
      for (int i = 0; i < C::NA; ++i) {
        const int k = k0 + Mp::kk(tid, i);
        r[i] = (m < p.M && k < p.K) ? base[(long)m * p.sam + (long)k * p.sak] : 0.f;
      }
    }
  }
  __device__ void store(const float (&r)[C::NA], float* t) {
    if (K_CONTIG) igemm::store_k_major<C::BM, C::KC, C::LDA>(r, t, tid);
    else igemm::store_mn_major<C::BM, C::KC, C::LDA>(r, t, tid);
  }
};
template <class C, bool K_CONTIG>
struct GB {
  const float* base; int n0, tid;
  __device__ void init(const GemmP& p, int n0_, int tid_, int zb) { n0 = n0_; tid = tid_; base = p.B + (long)zb * p.sbb; }
  __device__ void load(const GemmP& p, int k0, float (&r)[C::NB]) {
    if (K_CONTIG) {
      using Mp = KMajor<C::BN, C::KC>;
#pragma unroll
      for (int i = 0; i < C::NB; ++i) {
        const int n = n0 + Mp::mn(tid, i), k = k0 + Mp::kk(tid, i);
        r[i] = (n < p.N && k < p.K) ? base[(long)k * p.sbk + (long)n * p.sbn] : 0.f;
      }
    } else {
      using Mp = MnMajor<C::BN, C::KC>;
      const int n = n0 + Mp::mn(tid);
#pragma unroll
      for (int i = 0; i < C::NB; ++i) {
        const int k = k0 + Mp::kk(tid, i);
        r[i] = (n < p.N && k < p.K) ? base[(long)k * p.sbk + (long)n * p.sbn] : 0.f;
      }
    }
  }
  __device__ void store(const float (&r)[C::NB], float* t) {
    if (K_CONTIG) igemm::store_k_major<C::BN, C::KC, C::LDB>(r, t, tid);
    else igemm::store_mn_major<C::BN, C::KC, C::LDB>(r, t, tid);
  }
};
struct EpGemm {
  float* base;
  __device__ void init(const GemmP& p, int zb, int) { base = p.C + (long)zb * p.scb; }
  __device__ void col(const GemmP&, int) {}
  __device__ void put(const GemmP& p, int m, int n, float acc) {
    float v = p.alpha * acc;
    if (p.bias_mode == 1) v += p.bias[m];
    else if (p.bias_mode == 2) v += p.bias[n];
    float* c = base + (long)m * p.scm + (long)n * p.scn;
    *c = (p.beta != 0.f ? p.beta * *c : 0.f) + v;
  }
};

// ---- host-side helpers ----------------------------------------------------------------------------------------
// 128x128 tiles when they still give every CU work; 64x64 otherwise (small maps / small batches).
inline bool use_big_tile(int M, long N, int z) {
  const long t = (long)stk_cdiv(M, 128) * stk_cdiv(N, 128) * z;
  return M >= 96 && N >= 96 && t >= 192;
}

template <class C, class P, class AL, class BL, class EP>
int launch(const P& p, int M, long Nl, int K, int k_per_split, int splits, int batch, hipStream_t s) {
  if (Nl > 0x7fffffffL) return STK_EUNSUPPORTED;
  const int N = (int)Nl;
  const int tm = stk_cdiv(M, C::BM), tn = stk_cdiv(N, C::BN);
  dim3 grid((unsigned)(tm * tn), (unsigned)splits, (unsigned)batch);
  hipLaunchKernelGGL((igemm::kernel<C, P, AL, BL, EP>), grid, dim3(256), 0, s, p, M, N, K, tm, tn, k_per_split);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

inline int fill_common(ConvP& p, int N, int H, int W, int C1, int C2, int Cout, int OH, int OW, int KH, int KW,
                       int stride, int pad) {
  if (N <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || OH <= 0 || OW <= 0) return STK_EINVAL;
  if (!((KH == 3 && KW == 3) || (KH == 1 && KW == 1))) return STK_EUNSUPPORTED;
  if (stride != 1 && stride != 2) return STK_EUNSUPPORTED;
  p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.Cin = C1 + C2; p.Cout = Cout; p.OH = OH; p.OW = OW;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.sshift = stride == 2 ? 1 : 0;
  p.HW = H * W; p.OHW = OH * OW; p.taps = KH * KW;
  return STK_OK;
}

struct WgradPlan { int big; int splits; int k_per_split; long slab; };
inline WgradPlan wgrad_plan(int Cin, int N, int Cout, int OH, int OW, int KH, int KW) {
  WgradPlan q;
  const int taps = KH * KW;
  const long K = (long)N * OH * OW;
  q.big = (Cout >= 96 && Cin >= 96) ? 1 : 0;
  const int T = q.big ? 128 : 64;
  const long tiles = (long)stk_cdiv(Cout, T) * stk_cdiv(Cin, T) * taps;
  const long chunks = (K + 31) / 32;
  long splits = (512 + tiles - 1) / tiles;
  if (splits > chunks / 4) splits = chunks / 4;
  if (splits < 1) splits = 1;
  const long cps = (chunks + splits - 1) / splits;
  q.k_per_split = (int)(cps * 32);
  q.splits = (int)((K + q.k_per_split - 1) / q.k_per_split);
  q.slab = (long)Cout * Cin * taps;
  return q;
}

}  // namespace

extern "C" {

int stk_conv2d_fwd_f32(const float* x1, int C1, const float* x2, int C2, const float* w, int w_layout,
                       const float* bias, const float* temb, int temb_stride, const float* res, float out_div,
                       float* y, int N, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad,
                       void* stream) {
  if (!x1 || !w || !y || (C2 > 0 && !x2) || out_div == 0.f || (w_layout != 0 && w_layout != 1) ||
      (w_layout == 1 && (KH != 1 || KW != 1)))
    return STK_EINVAL;
  ConvP p = {};
  int rc = fill_common(p, N, H, W, C1, C2, Cout, OH, OW, KH, KW, stride, pad);
  if (rc) return rc;
  p.x1 = x1; p.x2 = C2 > 0 ? x2 : nullptr; p.w = w; p.w_layout = w_layout; p.bias = bias; p.temb = temb;
  p.temb_stride = temb_stride; p.res = res; p.inv_div = 1.f / out_div; p.use_div = out_div != 1.f; p.y = y;
  const int K = p.Cin * p.taps;
  const long Ng = (long)N * p.OHW;
  const bool big = use_big_tile(Cout, Ng, 1);
  hipStream_t s = (hipStream_t)stream;
  if (p.taps == 9) {
    p.sam = K; p.sak = 1;
    using CB = Cfg<128, 128, 36>; using CS = Cfg<64, 64, 36>;
    if (big) return launch<CB, ConvP, AFwd<CB, true>, BFwd<CB, 9>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
    return launch<CS, ConvP, AFwd<CS, true>, BFwd<CS, 9>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
  }
  using CB = Cfg<128, 128, 32>; using CS = Cfg<64, 64, 32>;
  if (w_layout == 0) {
    p.sam = K; p.sak = 1;
    if (big) return launch<CB, ConvP, AFwd<CB, true>, BFwd<CB, 1>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
    return launch<CS, ConvP, AFwd<CS, true>, BFwd<CS, 1>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
  }
  p.sam = 1; p.sak = Cout;   // NIN: w[ci][co]
  if (big) return launch<CB, ConvP, AFwd<CB, false>, BFwd<CB, 1>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
  return launch<CS, ConvP, AFwd<CS, false>, BFwd<CS, 1>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
}

int stk_conv2d_dgrad_f32(const float* dy, const float* w, int w_layout, float* dx1, int C1, float beta1, float* dx2,
                         int C2, float beta2, float alpha, int N, int H, int W, int Cout, int OH, int OW, int KH,
                         int KW, int stride, int pad, void* stream) {
  if (!dy || !w || (!dx1 && !dx2) || (w_layout != 0 && w_layout != 1) || (w_layout == 1 && (KH != 1 || KW != 1)))
    return STK_EINVAL;
  ConvP p = {};
  int rc = fill_common(p, N, H, W, C1, C2, Cout, OH, OW, KH, KW, stride, pad);
  if (rc) return rc;
  p.dy = dy; p.w = w; p.w_layout = w_layout; p.dx1 = dx1; p.dx2 = C2 > 0 ? dx2 : nullptr;
  p.beta1 = beta1; p.beta2 = beta2; p.alpha = alpha;
  const int Cin = p.Cin;
  const int K = Cout * p.taps;
  const long Ng = (long)N * p.HW;
  const bool big = use_big_tile(Cin, Ng, 1);
  hipStream_t s = (hipStream_t)stream;
  if (p.taps == 9) {
    using CB = Cfg<128, 128, 36>; using CS = Cfg<64, 64, 36>;
    if (big) return launch<CB, ConvP, ADgrad9<CB>, BDgrad<CB, 9>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
    return launch<CS, ConvP, ADgrad9<CS>, BDgrad<CS, 9>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
  }
  using CB = Cfg<128, 128, 32>; using CS = Cfg<64, 64, 32>;
  if (w_layout == 0) {       // A(m=ci,k=co) = w[co*Cin + ci]
    p.sam = 1; p.sak = Cin;
    if (big) return launch<CB, ConvP, ADgrad1<CB, false>, BDgrad<CB, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
    return launch<CS, ConvP, ADgrad1<CS, false>, BDgrad<CS, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
  }
  p.sam = Cout; p.sak = 1;   // NIN: A(m=ci,k=co) = w[ci*Cout + co]
  if (big) return launch<CB, ConvP, ADgrad1<CB, true>, BDgrad<CB, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
  return launch<CS, ConvP, ADgrad1<CS, true>, BDgrad<CS, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
}

long stk_conv2d_wgrad_ws_bytes(int C1, int C2, int N, int Cout, int OH, int OW, int KH, int KW) {
  const WgradPlan q = wgrad_plan(C1 + C2, N, Cout, OH, OW, KH, KW);
  return (long)q.splits * q.slab * 4 + 256;
}

int stk_conv2d_wgrad_f32(const float* x1, int C1, const float* x2, int C2, const float* dy, float* dw, int w_layout,
                         float alpha, float* ws, long ws_bytes, int N, int H, int W, int Cout, int OH, int OW, int KH,
                         int KW, int stride, int pad, void* stream) {
  if (!x1 || !dy || !dw || !ws || (C2 > 0 && !x2) || (w_layout != 0 && w_layout != 1) ||
      (w_layout == 1 && (KH != 1 || KW != 1)))
    return STK_EINVAL;
  ConvP p = {};
  int rc = fill_common(p, N, H, W, C1, C2, Cout, OH, OW, KH, KW, stride, pad);
  if (rc) return rc;
  const WgradPlan q = wgrad_plan(p.Cin, N, Cout, OH, OW, KH, KW);
  if (ws_bytes < (long)q.splits * q.slab * 4) return STK_EINVAL;
  p.x1 = x1; p.x2 = C2 > 0 ? x2 : nullptr; p.dy = dy; p.w_layout = w_layout; p.part = ws; p.part_stride = q.slab;
  const long Kl = (long)N * p.OHW;
  if (Kl > 0x7fffffffL) return STK_EUNSUPPORTED;
  const int K = (int)Kl;
  hipStream_t s = (hipStream_t)stream;
  using CB = Cfg<128, 128, 32>; using CS = Cfg<64, 64, 32>;
  if (q.big) rc = launch<CB, ConvP, AWgrad<CB>, BWgrad<CB>, EpWgrad>(p, Cout, p.Cin, K, q.k_per_split, q.splits, p.taps, s);
  else rc = launch<CS, ConvP, AWgrad<CS>, BWgrad<CS>, EpWgrad>(p, Cout, p.Cin, K, q.k_per_split, q.splits, p.taps, s);
  if (rc) return rc;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(stk_ew_grid(q.slab)), dim3(256), 0, s, ws, dw, q.slab, q.splits,
                     q.slab, alpha);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_gemm_f32(const float* A, long sam, long sak, long sab, const float* B, long sbk, long sbn, long sbb, float* C,
                 long scm, long scn, long scb, const float* bias, int bias_mode, int M, int N, int K, int batch,
                 float alpha, float beta, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || batch <= 0 || (bias_mode && !bias) || bias_mode < 0 ||
      bias_mode > 2)
    return STK_EINVAL;
  GemmP p;
  p.A = A; p.sam = sam; p.sak = sak; p.sab = sab; p.B = B; p.sbk = sbk; p.sbn = sbn; p.sbb = sbb;
  p.C = C; p.scm = scm; p.scn = scn; p.scb = scb; p.bias = bias; p.bias_mode = bias_mode;
  p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.beta = beta;
  const bool ak = sak == 1 || sam != 1;   // lanes along k unless m is the contiguous index
  const bool bk = sbk == 1 && sbn != 1;   // lanes along n unless only k is contiguous
  const bool big = use_big_tile(M, N, batch);
  hipStream_t s = (hipStream_t)stream;
  using CB = Cfg<128, 128, 32>; using CS = Cfg<64, 64, 32>;
#define STK_GEMM_CASE(AK, BK)                                                                              \
  if (ak == AK && bk == BK) {                                                                              \
    if (big) return launch<CB, GemmP, GA<CB, AK>, GB<CB, BK>, EpGemm>(p, M, N, K, K, 1, batch, s);         \
    return launch<CS, GemmP, GA<CS, AK>, GB<CS, BK>, EpGemm>(p, M, N, K, K, 1, batch, s);                  \
  }
  STK_GEMM_CASE(true, true)
  STK_GEMM_CASE(true, false)
  STK_GEMM_CASE(false, true)
  STK_GEMM_CASE(false, false)
#undef STK_GEMM_CASE
  return STK_EINVAL;
}

}  // extern "C"
