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
// instruction, shifted by the tap offset; halo / padding / stride-2 holes are zeros.
// The channel-concat of the up path (two sources) is just a source select on the channel index.
// Weights are read in their native layouts ([Cout,Cin,KH,KW] or NIN's [Cin,Cout]) by whichever lane
// mapping makes those reads contiguous; they are L2-resident (<= 4.7 MB per layer).
//
// wgrad reduces over batch*OH*OW (131072 for a 32x32 map at batch 128) into a small output, so it is
// split-K: each slice writes its partial tile to a scratch slab and a second kernel sums the slabs in a
// fixed order into dw (deterministic -- no float atomics).
#include "igemm.h"

namespace {

using igemm::Cfg;
using igemm::MnMajor;
using igemm::keep_if;
using igemm::strip_row;

// Global-address-space float: pointers rebuilt from integers must say so, or hipcc emits FLAT loads,
// which also count on lgkmcnt and would make the LDS waits of the MFMA loop wait for HBM.
typedef __attribute__((address_space(1))) float gfloat;

struct ConvP {
  const float* x1; const float* x2; int C1, C2;
  const float* w; int w_layout;
  const float* bias; const float* temb; int temb_stride; const float* res; float inv_div; int use_div;
  float* y;
  const float* dy; float* dx1; float* dx2; float beta1, beta2, alpha;
  float* part; long part_stride;
  int N, H, W, Cin, Cout, OH, OW, KH, KW, stride, pad, sshift;
  int HW, OHW, taps;
  int ohw_shift, ow_shift;    // log2 when OHW / OW are powers of two, else -1
};

// ---- loaders ---------------------------------------------------------------------------------------------
// Rules that shape them (measured: the first version, with one predicated load per element, compiled to an
// exec-mask branch plus an s_waitcnt per load and ran the fp32 MFMA pipe at ~50%):
//   * every load is UNCONDITIONAL from an always-valid address; validity goes into a per-thread bitmask and
//     the zeroing happens when the registers are written to LDS (after the MFMAs of the previous chunk), so
//     the 18 + 18 loads of a chunk issue back to back and stay in flight under the matrix work;
//   * activation bases are wave-uniform (SGPR) and per-lane offsets are 32-bit element offsets
//     (the host side rejects tensors of >= 2^31 elements): a load is `global_load_dword v, voff, s[base]`;
//   * K-contiguous operands (weights, GEMM rows) use a row-chunk mapping: 4 threads per row, each loading
//     KC/4 consecutive k at immediate offsets from one per-chunk address (merged into dwordx4 by hipcc).

// K-contiguous rows:  X(row, k) = base[row * ld + k],  row < R, k < Ktot.   Thread (r = tid/4, q = tid%4)
// loads k in [q*PERQ, (q+1)*PERQ) of rows r and (B == 128) r + 64.
template <int B, int KC, int LD>
struct RowChunk {
  static constexpr int PERQ = KC / 4, PASSES = B / 64, PER = PASSES * PERQ;
  static_assert(KC % 4 == 0 && B % 64 == 0 && PER <= 32, "bad RowChunk tile");
  const float* rowp[PASSES];     // start of this thread's row (row 0 of the operand when the row is out of range)
  unsigned rowbits;              // bit ps: row of pass ps is in range
  unsigned okm;                  // validity of the PER staged elements
  int q, tid, Ktot;
  __device__ __forceinline__ void init(const float* base, long ld, int row0, int R, int Ktot_, int tid_) {
    tid = tid_; q = tid & 3; Ktot = Ktot_; rowbits = 0; okm = 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int row = row0 + (tid >> 2) + 64 * ps;
      const bool ok = row < R;
      rowbits |= (ok ? 1u : 0u) << ps;
      rowp[ps] = base + (ok ? (long)row * ld : 0);
    }
  }
  // Ktot % PERQ == 0 is a precondition (the host dispatches ragged K to MnStrided), so a thread's PERQ
  // elements are valid or invalid together: no branch, PERQ loads at immediate offsets (-> dwordx4).
  __device__ __forceinline__ void load(int k0, float (&r)[PER]) {
    const int kb = k0 + q * PERQ;
    const bool kin = kb + PERQ <= Ktot;
    const int ko = kin ? kb : 0;
    okm = 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const float* s = rowp[ps] + ko;
#pragma unroll
      for (int j = 0; j < PERQ; ++j) r[ps * PERQ + j] = s[j];
      okm |= ((kin && ((rowbits >> ps) & 1u)) ? ((1u << PERQ) - 1u) : 0u) << (ps * PERQ);
    }
  }
  __device__ __forceinline__ void store(const float (&r)[PER], float* t) {
    if (okm == (PER == 32 ? 0xffffffffu : (1u << PER) - 1u)) {       // interior tile: nothing to zero
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
        for (int j = 0; j < PERQ; ++j) t[(q * PERQ + j) * LD + (tid >> 2) + 64 * ps] = r[ps * PERQ + j];
    } else {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
        for (int j = 0; j < PERQ; ++j)
          t[(q * PERQ + j) * LD + (tid >> 2) + 64 * ps] = keep_if(r[ps * PERQ + j], okm, ps * PERQ + j);
    }
  }
};

// MN-contiguous operand with a k stride:  X(mn, k) = base[mn * smn + k * sk]  (lanes along mn).
template <int B, int KC, int LD>
struct MnStrided {
  using Mp = MnMajor<B, KC>;
  static constexpr int PER = Mp::PER;
  const float* pm; long sk; int kg, tid, Ktot; bool ok;
  unsigned okm;
  __device__ __forceinline__ void init(const float* base, long smn, long sk_, int o0, int lim, int Ktot_, int tid_) {
    tid = tid_; sk = sk_; Ktot = Ktot_; okm = 0;
    const int mn = o0 + Mp::mn(tid);
    ok = mn < lim;
    pm = base + (ok ? (long)mn * smn : 0);
    kg = Mp::kgroup(tid) * PER;
  }
  __device__ __forceinline__ void load(int k0, float (&r)[PER]) {
    okm = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int k = k0 + kg + i;
      const bool v = ok && k < Ktot;
      r[i] = pm[v ? (long)k * sk : 0];
      okm |= (v ? 1u : 0u) << i;
    }
  }
  __device__ __forceinline__ void store(const float (&r)[PER], float* t) {
#pragma unroll
    for (int i = 0; i < PER; ++i) t[(kg + i) * LD + Mp::mn(tid)] = keep_if(r[i], okm, i);
  }
};

// ---- A loaders ----------------------------------------------------------------------------------------------
template <class C>
struct AFwdK {       // [Cout][Cin*taps] row-major weights: A(m=co, k) = w[co*K + k]
  RowChunk<C::BM, C::KC, C::LDA> rc;
  __device__ void init(const ConvP& p, int m0, int tid, int) { rc.init(p.w, (long)p.Cin * p.taps, m0, p.Cout, p.Cin * p.taps, tid); }
  __device__ void load(const ConvP&, int k0, float (&r)[C::NA]) { rc.load(k0, r); }
  __device__ void store(const float (&r)[C::NA], float* t) { rc.store(r, t); }
};
template <class C>
struct AFwdGen {     // same matrix, any K (e.g. Cin = 3): lanes along m, k strided by 1 -- uncoalesced but tiny
  MnStrided<C::BM, C::KC, C::LDA> ms;
  __device__ void init(const ConvP& p, int m0, int tid, int) { ms.init(p.w, (long)p.Cin * p.taps, 1, m0, p.Cout, p.Cin * p.taps, tid); }
  __device__ void load(const ConvP&, int k0, float (&r)[C::NA]) { ms.load(k0, r); }
  __device__ void store(const float (&r)[C::NA], float* t) { ms.store(r, t); }
};
template <class C>
struct AFwdNin {     // NIN w[Cin][Cout]: A(m=co, k=ci) = w[ci*Cout + co]  (m contiguous)
  MnStrided<C::BM, C::KC, C::LDA> ms;
  __device__ void init(const ConvP& p, int m0, int tid, int) { ms.init(p.w, 1, p.Cout, m0, p.Cout, p.Cin, tid); }
  __device__ void load(const ConvP&, int k0, float (&r)[C::NA]) { ms.load(k0, r); }
  __device__ void store(const float (&r)[C::NA], float* t) { ms.store(r, t); }
};
template <class C>
struct ADgradNin {   // NIN dgrad: A(m=ci, k=co) = w[ci*Cout + co]  (k contiguous)
  RowChunk<C::BM, C::KC, C::LDA> rc;
  __device__ void init(const ConvP& p, int m0, int tid, int) { rc.init(p.w, p.Cout, m0, p.Cin, p.Cout, tid); }
  __device__ void load(const ConvP&, int k0, float (&r)[C::NA]) { rc.load(k0, r); }
  __device__ void store(const float (&r)[C::NA], float* t) { rc.store(r, t); }
};
template <class C>
struct ADgradNinGen {   // NIN dgrad with Cout % 8 != 0
  MnStrided<C::BM, C::KC, C::LDA> ms;
  __device__ void init(const ConvP& p, int m0, int tid, int) { ms.init(p.w, p.Cout, 1, m0, p.Cin, p.Cout, tid); }
  __device__ void load(const ConvP&, int k0, float (&r)[C::NA]) { ms.load(k0, r); }
  __device__ void store(const float (&r)[C::NA], float* t) { ms.store(r, t); }
};
template <class C>
struct ADgrad1 {     // 1x1 Conv2d dgrad: A(m=ci, k=co) = w[co*Cin + ci]  (m contiguous)
  MnStrided<C::BM, C::KC, C::LDA> ms;
  __device__ void init(const ConvP& p, int m0, int tid, int) { ms.init(p.w, 1, p.Cin, m0, p.Cin, p.Cout, tid); }
  __device__ void load(const ConvP&, int k0, float (&r)[C::NA]) { ms.load(k0, r); }
  __device__ void store(const float (&r)[C::NA], float* t) { ms.store(r, t); }
};
// 3x3 dgrad on [Cout,Cin,3,3]:  A(m=ci, k=(co,tap)) = w[(co*Cin + ci)*9 + tap]: for a fixed (ci, co) the 9 taps
// are contiguous -> the row-chunk geometry with q = co within the chunk.
template <class C>
struct ADgrad9 {
  static_assert(C::KC == 36, "3x3 chunks are 4 channels x 9 taps");
  static constexpr int PASSES = C::BM / 64;
  int m0, tid; unsigned okp;
  __device__ void init(const ConvP&, int m0_, int tid_, int) { m0 = m0_; tid = tid_; okp = 0; }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NA]) {
    const int co = k0 / 9 + (tid & 3);
    okp = 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = m0 + (tid >> 2) + 64 * ps;
      const bool ok = m < p.Cin && co < p.Cout;
      const float* s = p.w + (ok ? ((long)co * p.Cin + m) * 9 : 0);
#pragma unroll
      for (int j = 0; j < 9; ++j) r[ps * 9 + j] = s[j];
      okp |= (ok ? 1u : 0u) << ps;
    }
  }
  __device__ void store(const float (&r)[C::NA], float* t) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
      for (int j = 0; j < 9; ++j)
        t[((tid & 3) * 9 + j) * C::LDA + (tid >> 2) + 64 * ps] = keep_if(r[ps * 9 + j], okp, ps);
  }
};

// ---- B loaders: activations, lanes along pixels -----------------------------------------------------------
// forward im2col:  B(k=(ci,tap), n=(b,oy,ox)) = x[b, ci, oy*s + kh - pad, ox*s + kw - pad]
template <class C, int TAPS, bool DUAL>
struct BFwd {
  using Mp = MnMajor<C::BN, C::KC>;
  static constexpr int NCH = C::NB / TAPS;     // channels per thread per chunk
  static_assert(Mp::PER % TAPS == 0 && C::NB <= 32, "k rows per thread must cover whole channels");
  int tb1, tb2;          // per-lane element offset of (image b, channel 0, pixel (iy0, ix0)) in x1 / x2
  int cig, tid; unsigned mask, okm;
  __device__ void init(const ConvP& p, int n0, int tid_, int) {
    tid = tid_;
    const int n = n0 + Mp::mn(tid);
    // channel offset of this thread's k rows inside a chunk; wave-uniform -> keep it in an SGPR
    cig = __builtin_amdgcn_readfirstlane(Mp::kgroup(tid) * (Mp::PER / TAPS));
    mask = 0; okm = 0; tb1 = 0; tb2 = 0;
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
      const int pix = iy0 * p.W + ix0;
      tb1 = b * p.C1 * p.HW + pix;
      tb2 = b * p.C2 * p.HW + pix;
    }
  }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NB]) {
    const int ci0 = k0 / TAPS + cig;          // scalar
    const int KW = TAPS == 9 ? 3 : 1;
    okm = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ci = ci0 + c;                 // scalar (k0 and cig are wave-uniform)
      const unsigned cm = ci < p.Cin ? mask : 0u;
      okm |= cm << (c * TAPS);
      const int cs = min(ci, p.Cin - 1);      // clamp so that the channel base below is always inside a tensor
      // Channel-plane base as a scalar address.  The source select is done on integers: a select between the
      // two tensor POINTERS made hipcc issue the load from both sources and select the value afterwards.
      const bool first = !DUAL || cs < p.C1;
      const uintptr_t tensor = first ? (uintptr_t)p.x1 : (uintptr_t)p.x2;
      const gfloat* plane = (const gfloat*)(tensor + (uintptr_t)(first ? cs : cs - p.C1) * (uintptr_t)p.HW * 4u);
      const int tb = first ? tb1 : tb2;
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
        r[c * TAPS + t] = plane[((cm >> t) & 1u) ? tb + (t / KW) * p.W + (t % KW) : 0];
    }
  }
  __device__ void store(const float (&r)[C::NB], float* t) {
    const int kg = Mp::kgroup(tid) * Mp::PER, mn = Mp::mn(tid);
#pragma unroll
    for (int i = 0; i < C::NB; ++i) t[(kg + i) * C::LDB + mn] = keep_if(r[i], okm, i);
  }
};

// dgrad gather:  B(k=(co,tap), n=(b,y,x)) = dy[b, co, (y+pad-kh)/s, (x+pad-kw)/s]  where divisible & in range
template <class C, int TAPS>
struct BDgrad {
  using Mp = MnMajor<C::BN, C::KC>;
  static constexpr int NCH = C::NB / TAPS;
  static_assert(Mp::PER % TAPS == 0 && C::NB <= 32, "k rows per thread must cover whole channels");
  int tb, cig, tid, yp, xp; unsigned mask, okm;
  __device__ void init(const ConvP& p, int n0, int tid_, int) {
    tid = tid_;
    const int n = n0 + Mp::mn(tid);
    cig = __builtin_amdgcn_readfirstlane(Mp::kgroup(tid) * (Mp::PER / TAPS));
    mask = 0; okm = 0; tb = 0; yp = 0; xp = 0;
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
      tb = b * p.Cout * p.OHW;
    }
  }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NB]) {
    const int co0 = k0 / TAPS + cig;          // scalar
    const int KW = TAPS == 9 ? 3 : 1;
    okm = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int co = co0 + c;
      const unsigned cm = co < p.Cout ? mask : 0u;
      const int cb = tb + co * p.OHW;
      okm |= cm << (c * TAPS);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int off = cb + ((yp - t / KW) >> p.sshift) * p.OW + ((xp - t % KW) >> p.sshift);
        r[c * TAPS + t] = p.dy[((cm >> t) & 1u) ? off : 0];
      }
    }
  }
  __device__ void store(const float (&r)[C::NB], float* t) {
    const int kg = Mp::kgroup(tid) * Mp::PER, mn = Mp::mn(tid);
#pragma unroll
    for (int i = 0; i < C::NB; ++i) t[(kg + i) * C::LDB + mn] = keep_if(r[i], okm, i);
  }
};

// ---- wgrad loaders (K = pixels, lanes along pixels; tap = blockIdx.z) ------------------------------------
// KC = 32: kk = tid & 31 is this thread's pixel inside the chunk for every element, mn = tid/32 + 8 i.
// Pixel index -> (image, pixel in image) and (row, column); shifts when the sizes are powers of two.
__device__ __forceinline__ void split_pixel(const ConvP& p, int k, int& b, int& ohw) {
  if (p.ohw_shift >= 0) { b = k >> p.ohw_shift; ohw = k & (p.OHW - 1); }
  else { b = k / p.OHW; ohw = k - b * p.OHW; }
}
__device__ __forceinline__ void split_row(const ConvP& p, int ohw, int& oy, int& ox) {
  if (p.ow_shift >= 0) { oy = ohw >> p.ow_shift; ox = ohw & (p.OW - 1); }
  else { oy = ohw / p.OW; ox = ohw - oy * p.OW; }
}

template <class C>
struct AWgrad {       // A(m=co, k=pixel) = dy[b, co, ohw]
  static_assert(C::KC == 32 && C::NA <= 32, "wgrad uses 32-pixel chunks");
  int roff[C::NA];     // (clamped) row offsets m*OHW: chunk-invariant, so they are computed once
  int tid; unsigned okrows, okm;
  __device__ void init(const ConvP& p, int m0, int tid_, int) {
    tid = tid_; okm = 0; okrows = 0;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) {
      const int m = m0 + (tid >> 5) + 8 * i;
      okrows |= (m < p.Cout ? 1u : 0u) << i;
      roff[i] = min(m, p.Cout - 1) * p.OHW;
    }
  }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NA]) {
    const int k = k0 + (tid & 31);
    const bool kv = k < p.N * p.OHW;
    int b, ohw;
    split_pixel(p, kv ? k : 0, b, ohw);
    const int base = b * p.Cout * p.OHW + ohw;     // pixel 0 of image 0 when the pixel is out of range
    okm = kv ? okrows : 0u;
#pragma unroll
    for (int i = 0; i < C::NA; ++i) r[i] = p.dy[base + roff[i]];
  }
  __device__ void store(const float (&r)[C::NA], float* t) {
#pragma unroll
    for (int i = 0; i < C::NA; ++i) t[(tid & 31) * C::LDA + (tid >> 5) + 8 * i] = keep_if(r[i], okm, i);
  }
};
template <class C, bool DUAL>
struct BWgrad {       // B(k=pixel, n=ci) = x[b, ci, oy*s + kh - pad, ox*s + kw - pad]
  static_assert(C::KC == 32 && C::NB <= 32, "wgrad uses 32-pixel chunks");
  int roff[C::NB];     // (clamped) channel offsets ci*HW inside its source tensor
  int tid, kh, kw; unsigned okrows, okm, first;
  __device__ void init(const ConvP& p, int n0, int tid_, int zb) {
    tid = tid_; okm = 0; okrows = 0; first = 0;
    kh = zb / p.KW; kw = zb - kh * p.KW;
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
      const int ci = min(n0 + (tid >> 5) + 8 * i, p.Cin - 1);
      okrows |= (n0 + (tid >> 5) + 8 * i < p.Cin ? 1u : 0u) << i;
      const bool f = !DUAL || ci < p.C1;
      first |= (f ? 1u : 0u) << i;
      roff[i] = (f ? ci : ci - p.C1) * p.HW;
    }
  }
  __device__ void load(const ConvP& p, int k0, float (&r)[C::NB]) {
    const int k = k0 + (tid & 31);
    bool kv = k < p.N * p.OHW;
    int b, ohw, oy, ox;
    split_pixel(p, kv ? k : 0, b, ohw);
    split_row(p, ohw, oy, ox);
    const int iy = oy * p.stride + kh - p.pad, ix = ox * p.stride + kw - p.pad;
    kv = kv && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const int pix = kv ? iy * p.W + ix : 0;
    const int b1 = b * p.C1 * p.HW + pix, b2 = b * p.C2 * p.HW + pix;
    okm = kv ? okrows : 0u;
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
      if (DUAL) {
        const bool f = (first >> i) & 1u;
        const uintptr_t tensor = f ? (uintptr_t)p.x1 : (uintptr_t)p.x2;
        r[i] = ((const gfloat*)tensor)[(f ? b1 : b2) + roff[i]];
      } else {
        r[i] = p.x1[b1 + roff[i]];
      }
    }
  }
  __device__ void store(const float (&r)[C::NB], float* t) {
#pragma unroll
    for (int i = 0; i < C::NB; ++i) t[(tid & 31) * C::LDB + (tid >> 5) + 8 * i] = keep_if(r[i], okm, i);
  }
};

// ---- epilogues: one 16-row strip of one output column per call, loads batched ------------------------------
struct EpFwd {        // y = (acc + bias + temb + res) * inv_div
  int b; int col_off;
  // Tile-level staging of the per-row addends (split kernels; conv_x2.h / conv_pl.h): a 128-row tile needs 128 bias
  // values and 128 time-embedding values per image it touches.  Fetched into registers BEFORE the main loop (their
  // latency hides under it), parked in LDS after it, read back with ds_read_b128 -- the epilogue itself then waits
  // for nothing but the residual loads, issued 16 at a time.  (The first version loaded bias / temb / res per group of
  // four rows: sixteen dependent round trips per wave, ~20 us of a 150 us launch at K = 1152.)
  static constexpr int TEMB_IMGS = 9;                 // 128-pixel tile over maps of >= 16 pixels
  float rb; float rt[5]; int m0 = 0, b0 = 0, nimg = 0, tb = 0;
  const float* sb = nullptr; const float* stm = nullptr;     // null: not staged (the f32-input and bf16 kernels)
  __device__ void preload(const ConvP& p, int m0_, int n0, int tn, int M, int Nn, int tid) {
    m0 = m0_; sb = nullptr; stm = nullptr;
    b0 = n0 / p.OHW;
    nimg = min(n0 + tn - 1, Nn - 1) / p.OHW - b0 + 1;
    rb = (p.bias && tid < 128 && m0 + tid < M) ? p.bias[m0 + tid] : 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int idx = tid + 256 * k, img = idx >> 7, r = idx & 127;
      rt[k] = (p.temb && nimg <= TEMB_IMGS && img < nimg && m0 + r < M) ? p.temb[(long)(b0 + img) * p.temb_stride + m0 + r] : 0.f;
    }
  }
  __device__ void stage(unsigned char* lds, int tid) {      // all waves are past their last operand read when they arrive
    __syncthreads();
    float* s = reinterpret_cast<float*>(lds);
    if (tid < 128) s[tid] = rb;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int idx = tid + 256 * k;
      if (idx < TEMB_IMGS * 128) s[128 + idx] = rt[k];
    }
    __syncthreads();
    sb = s; stm = s + 128;
  }
  __device__ void init(const ConvP&, int, int) {}
  __device__ void finish(const ConvP&, unsigned char*, int) {}
  __device__ void col(const ConvP& p, int n) {
    b = n / p.OHW;
    col_off = b * p.Cout * p.OHW + (n - b * p.OHW);
    tb = (b - b0) * 128;
  }
  __device__ void strip(const ConvP& p, int mbase, int M, bool nok, int, const floatx16& acc) {
    if (sb && (!p.temb || nimg <= TEMB_IMGS)) {      // staged addends (wave-uniform branch)
      const int ml = mbase - m0;
      float rv[16]; int idx[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mbase + strip_row(e);
        const bool ok = nok && m < M;
        idx[e] = ok ? col_off + m * p.OHW : -1;
        rv[e] = (p.res && ok) ? p.res[idx[e]] : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bb = *reinterpret_cast<const float4*>(sb + ml + 8 * g);
        const float4 tt = *reinterpret_cast<const float4*>(stm + tb + ml + 8 * g);
        const float bv[4] = {bb.x, bb.y, bb.z, bb.w}, tv[4] = {tt.x, tt.y, tt.z, tt.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float v = acc[4 * g + u];
          if (p.bias) v += bv[u];
          if (p.temb) v += tv[u];
          if (p.res) v += rv[4 * g + u];
          if (p.use_div) v *= p.inv_div;
          if (idx[4 * g + u] >= 0) p.y[idx[4 * g + u]] = v;
        }
      }
      return;
    }
    // 4 rows at a time: up to 12 independent loads in flight, and only a dozen live registers on top of
    // the accumulators (a 16-row batch pushed the whole kernel to 249 VGPRs = 2 waves/SIMD).
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float bv[4], tv[4], rv[4];
      int idx[4]; bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = mbase + strip_row(4 * g + u);
        ok[u] = nok && m < M;
        idx[u] = ok[u] ? col_off + m * p.OHW : 0;
        const int ms = ok[u] ? m : 0;
        bv[u] = p.bias ? p.bias[ms] : 0.f;
        tv[u] = p.temb ? p.temb[ok[u] ? (long)b * p.temb_stride + m : 0] : 0.f;
        rv[u] = p.res ? p.res[idx[u]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v = acc[4 * g + u];
        if (p.bias) v += bv[u];
        if (p.temb) v += tv[u];
        if (p.res) v += rv[u];
        if (p.use_div) v *= p.inv_div;
        if (ok[u]) p.y[idx[u]] = v;
      }
    }
  }
};
struct EpDgrad {      // dx{1,2} = beta*dx + alpha*acc, rows routed to the two sources of the concat
  int b, hw;
  __device__ void preload(const ConvP&, int, int, int, int, int, int) {}
  __device__ void stage(unsigned char*, int) {}
  __device__ void init(const ConvP&, int, int) {}
  __device__ void col(const ConvP& p, int n) { b = n / p.HW; hw = n - b * p.HW; }
  __device__ void strip(const ConvP& p, int mbase, int M, bool nok, int, const floatx16& acc) {
    // four elements at a time (the rows of one 4-row group of the strip): 16 pointers + 16 old values live at once kept every kernel with
    // this epilogue at 155 registers where the forward's has 126
    const bool acc1 = p.beta1 != 0.f, acc2 = p.beta2 != 0.f;
#pragma unroll
    for (int e0 = 0; e0 < 16; e0 += 4) {
      float* d[4]; float old[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = mbase + strip_row(e0 + u);
        float* q = nullptr;
        if (nok && m < M) {
          if (m < p.C1) { if (p.dx1) q = p.dx1 + ((long)b * p.C1 + m) * p.HW + hw; }
          else if (p.dx2) q = p.dx2 + ((long)b * p.C2 + (m - p.C1)) * p.HW + hw;
        }
        d[u] = q;
        old[u] = 0.f;
        if (acc1 || acc2) {
          const float beta = m < p.C1 ? p.beta1 : p.beta2;
          if (q && beta != 0.f) old[u] = beta * *q;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (d[u]) *d[u] = old[u] + p.alpha * acc[e0 + u];
    }
  }
  __device__ void finish(const ConvP&, unsigned char*, int) {}
};
struct EpWgrad {      // partial slab of split zs as [tap][Cout][Cin] (coalesced); the reduce kernel re-lays it out
  float* slab; int tap;
  __device__ void init(const ConvP& p, int zb, int zs) { slab = p.part + (long)zs * p.part_stride; tap = zb; }
  __device__ void col(const ConvP&, int) {}
  __device__ void strip(const ConvP& p, int mbase, int M, bool nok, int n, const floatx16& acc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = mbase + strip_row(e);
      if (nok && m < M) {
        slab[((long)tap * p.Cout + m) * p.Cin + n] = acc[e];     // [tap][co][ci]: lanes (= ci) contiguous
      }
    }
  }
};

// Forward / dgrad split over K (small maps): partial tile of split zs as [M][N] (lanes = pixels, contiguous)
struct EpSlab {
  float* slab; int Nn;
  __device__ void preload(const ConvP&, int, int, int, int, int, int) {}
  __device__ void stage(unsigned char*, int) {}
  __device__ void init(const ConvP& p, int, int zs) { slab = p.part + (long)zs * p.part_stride; Nn = p.N * p.HW; }
  __device__ void finish(const ConvP&, unsigned char*, int) {}
  __device__ void col(const ConvP&, int) {}
  __device__ void strip(const ConvP&, int mbase, int M, bool nok, int n, const floatx16& acc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = mbase + strip_row(e);
      if (nok && m < M) slab[(long)m * Nn + n] = acc[e];
    }
  }
};
// sum of the split slabs in a fixed order, then the same epilogue arithmetic as EpFwd / EpDgrad.
// VEC = 4 (H W % 4 == 0, 16-byte aligned tensors): four consecutive pixels of one image per thread -- 16-byte loads of the
// `splits` slabs and of the residual, one 16-byte store (round 4: the scalar form spent 9 us on 33 MB).
template <int VEC>
__global__ __launch_bounds__(256) void slab_fwd_kernel(ConvP p, int splits, int M, int Nn) {
  const long total = (long)M * Nn / VEC, gstride = (long)gridDim.x * 256;
  for (long iv = (long)blockIdx.x * 256 + threadIdx.x; iv < total; iv += gstride) {
    const long i = iv * VEC;
    const int m = (int)(i / Nn), n = (int)(i - (long)m * Nn);
    float v[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) v[u] = 0.f;
    for (int z = 0; z < splits; ++z) {
      const float* q = p.part + (long)z * p.part_stride + i;
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(q);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      } else {
        v[0] += q[0];
      }
    }
    const int b = n / p.OHW;
    const long idx = ((long)b * p.Cout + m) * p.OHW + (n - b * p.OHW);
    float r[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) r[u] = 0.f;
    if (p.res) {
      if constexpr (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p.res + idx); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
      else r[0] = p.res[idx];
    }
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      // same order of additions as the scalar epilogue: acc + bias + temb + res, then the division
      float w = v[u];
      if (p.bias) w += p.bias[m];
      if (p.temb) w += p.temb[(long)b * p.temb_stride + m];
      if (p.res) w += r[u];
      if (p.use_div) w *= p.inv_div;
      v[u] = w;
    }
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(p.y + idx) = make_float4(v[0], v[1], v[2], v[3]);
    else p.y[idx] = v[0];
  }
}
template <int VEC>
__global__ __launch_bounds__(256) void slab_dgrad_kernel(ConvP p, int splits, int M, int Nn) {
  const long total = (long)M * Nn / VEC, gstride = (long)gridDim.x * 256;
  for (long iv = (long)blockIdx.x * 256 + threadIdx.x; iv < total; iv += gstride) {
    const long i = iv * VEC;
    const int m = (int)(i / Nn), n = (int)(i - (long)m * Nn);
    float v[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) v[u] = 0.f;
    for (int z = 0; z < splits; ++z) {
      const float* q = p.part + (long)z * p.part_stride + i;
      if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(q);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      } else {
        v[0] += q[0];
      }
    }
    const int b = n / p.HW, hw = n - b * p.HW;
    float* d; float beta;
    if (m < p.C1) { d = p.dx1 ? p.dx1 + ((long)b * p.C1 + m) * p.HW + hw : nullptr; beta = p.beta1; }
    else { d = p.dx2 ? p.dx2 + ((long)b * p.C2 + (m - p.C1)) * p.HW + hw : nullptr; beta = p.beta2; }
    if (!d) continue;
    if constexpr (VEC == 4) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (beta != 0.f) { const float4 t = *reinterpret_cast<const float4*>(d); o = make_float4(beta * t.x, beta * t.y, beta * t.z, beta * t.w); }
      *reinterpret_cast<float4*>(d) = make_float4(o.x + p.alpha * v[0], o.y + p.alpha * v[1], o.z + p.alpha * v[2], o.w + p.alpha * v[3]);
    } else {
      *d = (beta != 0.f ? beta * *d : 0.f) + p.alpha * v[0];
    }
  }
}
// launch of the slab sum: the 16-byte form when four consecutive pixels never leave an image / a tensor row
inline void launch_slab_sum(const ConvP& p, int splits, int M, long Ng, int dgrad, hipStream_t s) {
  const bool al = stk_aligned16(p.part) && p.part_stride % 4 == 0 &&
                  (dgrad ? (p.HW % 4 == 0 && (!p.dx1 || stk_aligned16(p.dx1)) && (!p.dx2 || stk_aligned16(p.dx2)))
                         : (p.OHW % 4 == 0 && stk_aligned16(p.y) && (!p.res || stk_aligned16(p.res))));
  if (al && Ng % 4 == 0) {
    const dim3 rgrid((unsigned)stk_ew_grid((long)M * Ng / 4));
    if (dgrad) hipLaunchKernelGGL(slab_dgrad_kernel<4>, rgrid, dim3(256), 0, s, p, splits, M, (int)Ng);
    else hipLaunchKernelGGL(slab_fwd_kernel<4>, rgrid, dim3(256), 0, s, p, splits, M, (int)Ng);
  } else {
    const dim3 rgrid((unsigned)stk_ew_grid((long)M * Ng));
    if (dgrad) hipLaunchKernelGGL(slab_dgrad_kernel<1>, rgrid, dim3(256), 0, s, p, splits, M, (int)Ng);
    else hipLaunchKernelGGL(slab_fwd_kernel<1>, rgrid, dim3(256), 0, s, p, splits, M, (int)Ng);
  }
}

#include "conv_x3.h"
#include "conv_thin.h"
#include "conv_x2.h"
#include "conv_pl.h"
#include "conv_x2d.h"
#include "conv_x2w.h"

// dw (in the weight's own layout) += alpha * sum over splits of slab[tap][co][ci]
// Threads walk the SLAB order four elements at a time, so the `splits` reads per element are 16-byte and coalesced
// and only the single read-modify-write of dw is scattered (stride `taps` floats).  Indexing by the dw order instead
// made every slab read a 9-way scatter and the kernel 2-3x slower than the partial sums' HBM time.
// Summation order over the splits is fixed (z ascending): results do not depend on the launch geometry.
__device__ __forceinline__ long splitk_dst(long j, int layout, long CC, int Cout, int Cin, int taps) {
  if (layout == 0) {             // j = tap*Cout*Cin + (co*Cin + ci)  ->  (co*Cin + ci)*taps + tap
    const int tap = (int)(j / CC);
    return (j - (long)tap * CC) * taps + tap;
  }
  const int co = (int)(j / Cin), ci = (int)(j - (long)co * Cin);      // NIN: j = co*Cin + ci  ->  ci*Cout + co
  return (long)ci * Cout + co;
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                            long n, int splits, long stride, float alpha, int layout,
                                                            int Cout, int Cin, int taps) {
  const long gstride = (long)gridDim.x * 256;
  const long CC = (long)Cout * Cin;
  const long n4 = ((n | stride) & 3) == 0 ? n / 4 : 0;                // float4 path needs 16-byte aligned slabs
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < n4; v += gstride) {
    const float4* src = reinterpret_cast<const float4*>(part) + v;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int z = 0; z < splits; ++z) {
      const float4 t = src[(long)z * (stride / 4)];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) dw[splitk_dst(4 * v + e, layout, CC, Cout, Cin, taps)] += alpha * a4[e];
  }
  for (long j = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; j < n; j += gstride) {
    float sum = 0.f;
    for (int z = 0; z < splits; ++z) sum += part[(long)z * stride + j];
    dw[splitk_dst(j, layout, CC, Cout, Cin, taps)] += alpha * sum;
  }
}

// The same sum for 3x3 layers in the weight's own layout (w_layout 0), with COALESCED read-modify-writes of dw (round 4).
// The kernel above reads the slabs in 16-byte runs but scatters its dw updates with a stride of nine floats: a wave touches
// 72 cache lines for 256 values, nine times over per line.  Here a workgroup owns 256 consecutive (co, ci) pairs: thread t sums
// its pair's nine taps over the splits (4-byte loads, contiguous over the workgroup: 1 KB per tap and slab), the 2304 sums
// meet in LDS in dw order, and the update of dw is nine fully coalesced 1 KB read-modify-writes.  Same summation order per
// element (z ascending): bit-identical results.
__global__ __launch_bounds__(256) void splitk_reduce9_kernel(const float* __restrict__ part, float* __restrict__ dw, long CC,
                                                             int splits, long stride, float alpha) {
  __shared__ float s[256 * 9];
  const long q0 = (long)blockIdx.x * 256;
  const long q = q0 + threadIdx.x;
  const bool live = q < CC;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  const float* src = part + (live ? q : 0);
  int z = 0;
  for (; z + 4 <= splits; z += 4) {          // 36 independent loads in flight, added in z order
    float v[4][9];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 9; ++t) v[u][t] = src[(long)(z + u) * stride + (long)t * CC];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] += v[u][t];
  }
  for (; z < splits; ++z)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] += src[(long)z * stride + (long)t * CC];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[threadIdx.x * 9 + t] = alpha * acc[t];
  __syncthreads();
  const long base = q0 * 9, end = CC * 9;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i < end) dw[i] += s[k * 256 + threadIdx.x];
  }
}

// ---- 3x3 / stride-1 / pad-1 weight gradient: all nine taps per workgroup ------------------------------------
// The per-tap GEMM above re-reads dy and x once per tap and per tile and is bound by L2/Infinity-Cache
// bandwidth (measured ~70 TFLOP/s).  Here one workgroup owns a 128(co) x 32(ci) x 9(taps) block of dw:
// per 32-pixel chunk it stages dy[128 co][32 px] and ONE halo tile x[32 ci][(rows+2) x (cols+2)] in LDS and
// feeds all nine taps from shifted views of that tile: 144 MFMAs per wave per chunk against 29 loads per
// thread (the per-tap kernel: 64 MFMAs against 32 loads), i.e. ~4.5x fewer operand bytes per FLOP.
//   A operand  As[px][co]   (k-major, pitch 129)             a = As[2ks + (lane>>5)][32*wave + (lane&31)]
//   B operand  Xs[ci][tile] (pitch odd -> conflict-free)     b_tap = Xs[lane&31][loc(2ks) + (lane>>5) + kh*TW + kw]
// COLS = pixels of a chunk that share an image row: 32 (W % 32 == 0), 16 (W == 16) or 8 (W == 8); a chunk is
// 32/COLS full rows of one image, so it never straddles images (requires OH*OW % 32 == 0).
template <int COLS>
__global__ __launch_bounds__(256) void wgrad9_kernel(ConvP p, int tiles_m, int tiles_n, int k_per_split) {
  constexpr int ROWS = 32 / COLS, TW = COLS + 2, TH = ROWS + 2, TSZ = TH * TW;
  constexpr int XS = TSZ | 1;
  constexpr int LDA = 129;
  constexpr int NX = (32 * TSZ + 255) / 256;
  constexpr int NA = 16;
  __shared__ float lds[32 * LDA + 32 * XS];
  float* As = lds;
  float* Xs = lds + 32 * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fk = lane >> 5, fc = lane & 31;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = id % tiles_n;               // ci tiles fastest: neighbours share the dy panel
  const int rest = id / tiles_n;
  const int tm = rest % tiles_m, zs = rest / tiles_m;
  const int m0 = tm * 128, n0 = tn * 32;
  const int K = p.N * p.OHW;
  const int k_begin = zs * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);

  // dy rows of this thread: m = mrow + 8 i (row offsets are recomputed per chunk, not held in registers)
  const int mrow = m0 + (tid >> 5), cout_hi = p.Cout - 1;
  unsigned okrowsA = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) okrowsA |= (mrow + 8 * i < p.Cout ? 1u : 0u) << i;
  // x-tile element e = tid + 256 i  ->  (ci = e / TSZ, lr, lc); recomputed per chunk from compile-time divisors
  // (a few VALU ops under 9k cycles of MFMA) instead of being held in 39 registers, which keeps the kernel
  // at two waves per SIMD.
  const int cin_hi = p.Cin - 1;
  floatx16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  float ra[NA], rx[NX];
  unsigned okmA = 0, okmX = 0;
  auto load = [&](int k0) {
    // dy: this thread's pixel is k0 + (tid & 31)
    const int k = k0 + (tid & 31);
    const bool kv = k < K;
    int b, ohw;
    split_pixel(p, kv ? k : 0, b, ohw);
    const int abase = b * p.Cout * p.OHW + ohw;
    okmA = kv ? okrowsA : 0u;
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = p.dy[abase + min(mrow + 8 * i, cout_hi) * p.OHW];
    // x halo tile of the chunk (scalar origin: k0 is chunk-aligned and the chunk lies in one image)
    int b0, ohw0, oy0, ox0;
    split_pixel(p, k0, b0, ohw0);
    split_row(p, ohw0, oy0, ox0);
    const int xbase = b0 * p.Cin * p.HW + (oy0 - 1) * p.W + (ox0 - 1);
    okmX = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i;
      const int ci = e / TSZ, rem = e - ci * TSZ;
      const int lr = rem / TW, lc = rem - lr * TW;
      const int iy = oy0 - 1 + lr, ix = ox0 - 1 + lc;
      const bool ok = e < 32 * TSZ && n0 + ci < p.Cin && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      rx[i] = p.x1[ok ? xbase + min(n0 + ci, cin_hi) * p.HW + lr * p.W + lc : 0];
      okmX |= (ok ? 1u : 0u) << i;
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) As[(tid & 31) * LDA + (tid >> 5) + 8 * i] = keep_if(ra[i], okmA, i);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + 256 * i;
      const int ci = e / TSZ, rem = e - ci * TSZ;
      if (e < 32 * TSZ) Xs[ci * XS + rem] = keep_if(rx[i], okmX, i);
    }
  };

  if (k_begin < k_end) load(k_begin);
  const float* arow = As + fk * LDA + wid * 32 + fc;
  const float* xrow = Xs + fc * XS + fk;
  for (int k0 = k_begin; k0 < k_end; k0 += 32) {
    store();
    __syncthreads();
    if (k0 + 32 < k_end) load(k0 + 32);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float a = arow[2 * ks * LDA];
      constexpr int dummy = 0; (void)dummy;
      const int loc = ((2 * ks) / COLS) * TW + ((2 * ks) % COLS);
      float bv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) bv[t] = xrow[loc + (t / 3) * TW + (t % 3)];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }

  // partial slab [tap][Cout][Cin] of split zs: lanes (= ci) contiguous
  float* slab = p.part + (long)zs * p.part_stride;
  const int n = n0 + fc;
  if (n < p.Cin) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wid * 32 + strip_row(e) + 4 * fk;
        if (m < p.Cout) slab[((long)t * p.Cout + m) * p.Cin + n] = acc[t][e];
      }
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
struct GA {           // A(m,k): K_CONTIG -> rows of contiguous k (sak == 1); else lanes along m with a k stride
  RowChunk<C::BM, C::KC, C::LDA> rc;
  MnStrided<C::BM, C::KC, C::LDA> ms;
  __device__ void init(const GemmP& p, int m0, int tid, int zb) {
    const float* base = p.A + (long)zb * p.sab;
    if (K_CONTIG) rc.init(base, p.sam, m0, p.M, p.K, tid);
    else ms.init(base, p.sam, p.sak, m0, p.M, p.K, tid);
  }
  __device__ void load(const GemmP&, int k0, float (&r)[C::NA]) {
    if (K_CONTIG) rc.load(k0, r); else ms.load(k0, r);
  }
  __device__ void store(const float (&r)[C::NA], float* t) {
    if (K_CONTIG) rc.store(r, t); else ms.store(r, t);
  }
};
template <class C, bool K_CONTIG>
struct GB {           // B(k,n): K_CONTIG -> columns stored as rows of contiguous k (sbk == 1); else lanes along n
  RowChunk<C::BN, C::KC, C::LDB> rc;
  MnStrided<C::BN, C::KC, C::LDB> ms;
  __device__ void init(const GemmP& p, int n0, int tid, int zb) {
    const float* base = p.B + (long)zb * p.sbb;
    if (K_CONTIG) rc.init(base, p.sbn, n0, p.N, p.K, tid);
    else ms.init(base, p.sbn, p.sbk, n0, p.N, p.K, tid);
  }
  __device__ void load(const GemmP&, int k0, float (&r)[C::NB]) {
    if (K_CONTIG) rc.load(k0, r); else ms.load(k0, r);
  }
  __device__ void store(const float (&r)[C::NB], float* t) {
    if (K_CONTIG) rc.store(r, t); else ms.store(r, t);
  }
};
struct EpGemm {
  float* base;
  __device__ void init(const GemmP& p, int zb, int) { base = p.C + (long)zb * p.scb; }
  __device__ void col(const GemmP&, int) {}
  __device__ void strip(const GemmP& p, int mbase, int M, bool nok, int n, const floatx16& acc) {
    float old[16];
    if (p.beta != 0.f) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mbase + strip_row(e);
        const bool ok = nok && m < M;
        old[e] = base[ok ? (long)m * p.scm + (long)n * p.scn : 0];
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = mbase + strip_row(e);
      if (nok && m < M) {
        float v = p.alpha * acc[e];
        if (p.bias_mode == 1) v += p.bias[m];
        else if (p.bias_mode == 2) v += p.bias[n];
        base[(long)m * p.scm + (long)n * p.scn] = (p.beta != 0.f ? p.beta * old[e] : 0.f) + v;
      }
    }
  }
};

// ---- host-side helpers ----------------------------------------------------------------------------------------
// 128x128 tiles when they still give every CU work; 64x64 otherwise (small maps / small batches).
inline bool use_big_tile(int M, long N, int z) {
  const long t = (long)stk_cdiv(M, 128) * stk_cdiv(N, 128) * z;
  return M >= 96 && N >= 96 && t >= 192;
}

template <class C, class P, class AL, class BL, class EP>
int launch(const P& p, int M, long Nl, int K, int k_per_split, int splits, int batch, hipStream_t s,
           bool flat = false) {
  if (Nl > 0x7fffffffL) return STK_EUNSUPPORTED;
  const int N = (int)Nl;
  const int tm = stk_cdiv(M, C::BM), tn = stk_cdiv(N, C::BN);
  if (flat) {      // batch (= taps) fastest, then tiles, then splits, all in grid.x
    const long total = (long)tm * tn * splits * batch;
    if (total > 0x7fffffffL) return STK_EUNSUPPORTED;
    hipLaunchKernelGGL((igemm::kernel<C, P, AL, BL, EP>), dim3((unsigned)total), dim3(256), 0, s, p, M, N, K, tm, tn,
                       k_per_split, batch);
  } else {
    dim3 grid((unsigned)(tm * tn), (unsigned)splits, (unsigned)batch);
    hipLaunchKernelGGL((igemm::kernel<C, P, AL, BL, EP>), grid, dim3(256), 0, s, p, M, N, K, tm, tn, k_per_split, 0);
  }
  STK_CHECK_LAUNCH();
  return STK_OK;
}

inline int fill_common(ConvP& p, int N, int H, int W, int C1, int C2, int Cout, int OH, int OW, int KH, int KW,
                       int stride, int pad) {
  if (N <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || OH <= 0 || OW <= 0) return STK_EINVAL;
  if (!((KH == 3 && KW == 3) || (KH == 1 && KW == 1))) return STK_EUNSUPPORTED;
  if (stride != 1 && stride != 2) return STK_EUNSUPPORTED;
  // 32-bit element offsets inside the kernels
  const long lim = 0x7fffffffL;
  if ((long)N * (C1 > C2 ? C1 : C2) * H * W >= lim || (long)N * Cout * OH * OW >= lim) return STK_EUNSUPPORTED;
  p.N = N; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.Cin = C1 + C2; p.Cout = Cout; p.OH = OH; p.OW = OW;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.sshift = stride == 2 ? 1 : 0;
  p.HW = H * W; p.OHW = OH * OW; p.taps = KH * KW;
  p.ohw_shift = -1; p.ow_shift = -1;
  for (int sft = 0; sft < 31; ++sft) {
    if ((1 << sft) == p.OHW) p.ohw_shift = sft;
    if ((1 << sft) == p.OW) p.ow_shift = sft;
  }
  return STK_OK;
}

// The bf16 three-way-split kernel (conv_x3.h) takes 3x3 / stride 1 / pad 1 and 1x1 / stride 1 layers whose channel
// count is a multiple of the 32-wide k chunk (and, for a concat input, whose first source is too).  With at least
// 192 tiles of 128 x 128 a workgroup owns a whole output tile.  Smaller problems (the 8x8 and 4x4 maps: 128 / 32
// tiles at batch 128, where one workgroup per CU has nothing to overlap with) are split over K into partial slabs
// that a second kernel sums in a fixed order and finishes with the usual epilogue.
struct X3Plan { int ok; int splits; int chunks_per_split; long slab; int t64; int gps; };
inline X3Plan x3_plan(const ConvP& p, int Kc, int S1, int S2, int M, long Ng) {
  X3Plan r = {0, 1, 0, 0, 0, 0};
  const long big = (long)p.N * p.HW * 4 * (S1 > S2 ? S1 : S2);       // buffer loads: 32-bit byte offsets, bit 31 = dead lane
  const bool geom = (p.taps == 9 && p.pad == 1) || (p.taps == 1 && p.pad == 0);
  if (!(big < 0x7fffffffL && geom && p.stride == 1 && p.OH == p.H && p.OW == p.W && Kc % 32 == 0 &&
        (S2 == 0 || S1 % 32 == 0) && M >= 96 && Ng <= 0x7fffffffL))
    return r;
  const long tiles = (long)stk_cdiv(M, 128) * stk_cdiv(Ng, 128);
  const int nch = p.taps * (Kc / 32);
  r.chunks_per_split = nch;
  if (tiles >= 192) { r.ok = 1; return r; }
  // STK_KSPLIT_WGS: A-B knob of the split (workgroups to fill); at least `minch` chunks per workgroup
  static const long target = [] { const char* e = getenv("STK_KSPLIT_WGS"); return e && atol(e) > 0 ? atol(e) : 512L; }();
  // (round 4 sweep, profiles/r04_ksplit_sweep.txt: >= 12 chunks per workgroup -- 256 -> 256 at 4x4, batch 128: 12 splits 27.4 us,
  // 6 splits 23.8; 512 -> 256 at 8x8: 73.9 -> 71.9; the 8x8 256 -> 256 layers keep their 4 splits either way)
  constexpr long minch = 12;
  long splits = target / tiles;
  // few chunks per workgroup: the prologue and the slab traffic dominate (1x1 layers have Kc / 32 chunks in all: they keep 6)
  const long mc = p.taps == 9 ? minch : (minch < 6 ? minch : 6);
  if (splits > nch / mc) splits = nch / mc;
  // few tiles: the 3x3 layers still pay (4 tiles of a 256-channel 8x8 map at batch 4: 94 us on 16 workgroups of the
  // f32-input kernel); a 1x1 layer has too few chunks to split
  if ((tiles < 16 && p.taps != 9) || splits < 2) return r;
  r.chunks_per_split = (int)((nch + splits - 1) / splits);
  r.splits = (nch + r.chunks_per_split - 1) / r.chunks_per_split;
  r.slab = (long)M * Ng;
  r.ok = 1;
  return r;
}
// Plan of a PLANE-operand call (single source): the same plan (the 64 x 64-tile kernel of round 5 lost its in-step A/B and is retired)
inline X3Plan x3_plan_pl(const ConvP& p, int Kc, int M, long Ng) { return x3_plan(p, Kc, Kc, 0, M, Ng); }
inline long x3_ws_bytes(const X3Plan& r, int M, int Kc, int taps) {
  // [prepared weights of this call][|x| partial maxima, 2 x 256 floats][K-split slabs], each 256-byte aligned
  return r.ok ? x3::wp_bytes(M, Kc, taps) + 2048 + 1024 + (r.splits > 1 ? r.splits * r.slab * 4 : 0) : 0;
}
// dgrad = 1: rows are input channels, k output channels, taps flipped.  Every split kernel is the fp16 two-way split (3 MFMAs per
// fp32 product, conv_x2.h / conv_x2d.h / conv_x2w.h); the bf16 three-way-split kernels of round 1 (six MFMAs) were the fallback
// of a debugging switch until round 5 and are retired (DESIGN.md "Retired").

inline void x3_weight_strides(const ConvP& p, int dgrad, long& sm, long& sk) {
  if (p.w_layout == 0) { sm = dgrad ? p.taps : (long)p.Cin * p.taps; sk = dgrad ? (long)p.Cin * p.taps : p.taps; }
  else { sm = dgrad ? p.Cout : 1; sk = dgrad ? 1 : p.Cout; }          // NIN w[Cin][Cout]
}
// wp_ready: weights already prepared by stk_conv2d_wprep_batch (then ws only holds the K-split slabs)
template <class EP>
int launch_x3(ConvP p, const X3Plan& r, const float* s1, int S1, const float* s2, int S2, int M, long Ng, int dgrad,
              void* ws, hipStream_t s, const void* wp_ready = nullptr, float* amax = nullptr,
              const void* planes = nullptr, const float* planes_amax = nullptr, bool amax_valid = false) {
  x3::Src q;
  q.s1 = s1; q.s2 = S2 > 0 ? s2 : s1; q.S1 = S1; q.S2 = S2; q.Kc = S1 + S2; q.Mpad = x3::pad128(M); q.taps = p.taps;
  q.pl = static_cast<const unsigned char*>(planes); q.pl_stride = planes ? pl::plane_bytes(p.N, S1, p.HW) : 0;
  unsigned short* wp = reinterpret_cast<unsigned short*>(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* xpart = reinterpret_cast<float*>(((uintptr_t)wp + x3::wp_bytes(M, q.Kc, p.taps) + 255) & ~(uintptr_t)255);
  p.part = reinterpret_cast<float*>(((uintptr_t)(xpart + 2 * x2::NPART) + 255) & ~(uintptr_t)255);
  p.part_stride = r.slab;
  {
    // fp16 two-way split (conv_x2.h): |x| maxima of the activation operand(s), weights prepared here unless the caller did
    // with a caller-owned amax buffer (768 floats: |x1|, |x2|, |dy| partials) the maxima stay available to the layer's
    // weight gradient, which would otherwise repeat these passes
    if (amax) xpart = amax + (dgrad ? 2 * x2::NPART : 0);
    if (planes) {
      xpart = const_cast<float*>(planes_amax);      // the scale record the planes were written with
    } else if (!(amax_valid && amax)) {   // amax_valid: the caller's record already holds this operand's maxima (both sources)
      hipLaunchKernelGGL(x2::amax_partial_kernel, dim3(x2::NPART), dim3(x2::AMAX_THREADS), 0, s, s1, (long)p.N * S1 * p.HW, xpart);
      if (S2 > 0)
        hipLaunchKernelGGL(x2::amax_partial_kernel, dim3(x2::NPART), dim3(x2::AMAX_THREADS), 0, s, s2, (long)p.N * S2 * p.HW,
                           xpart + x2::NPART);
      STK_CHECK_LAUNCH();
    }
    const int nx = S2 > 0 ? 2 * x2::NPART : x2::NPART;
    if (wp_ready) {
      q.wp = static_cast<const unsigned short*>(wp_ready);
    } else {
      x2::WprepDesc one = {};
      one.w = p.w; one.wp = reinterpret_cast<unsigned char*>(wp); one.M = M; one.Kc = q.Kc; one.Mpad = q.Mpad;
      one.taps = p.taps; one.flip = dgrad;
      x3_weight_strides(p, dgrad, one.sm, one.sk);
      hipLaunchKernelGGL(x2::wamax_kernel, dim3(x2::WPART, 1), dim3(256), 0, s, nullptr, one);
      hipLaunchKernelGGL(x2::wprep_kernel, dim3((unsigned)stk_cdiv((long)q.Mpad * q.Kc, 256L)), dim3(256), 0, s, nullptr, one);
      STK_CHECK_LAUNCH();
      q.wp = wp;
    }
    const int tm = q.Mpad / 128, tn = stk_cdiv((int)Ng, 128), nch = p.taps * (q.Kc / x3::KC);
    const dim3 grid((unsigned)(tm * tn * r.splits));
#define STK_X2_LAUNCH(E, DUAL, TAPS)                                                                              \
  hipLaunchKernelGGL((x2::gemm_kernel<x2::ActLoader<DUAL, TAPS>, E>), grid, dim3(256), 0, s, p, q, M, (int)Ng, tm, tn, \
                     nch, r.chunks_per_split, xpart, nx)
#define STK_PL_LAUNCH(E, TAPS)                                                                                    \
  if (x2d::halo_ok(p, TAPS, r.splits)) {                                                                          \
    /* one halo tile of the activations per channel group serves the nine taps */                                 \
    if (p.W == 64)                                                                                                \
      hipLaunchKernelGGL((x2d::gemm_halo_kernel<64, E, 1>), grid, dim3(256), 0, s, p, q, M, (int)Ng, tm, tn, nch / 9, xpart, nx); \
    else if (p.W == 32)                                                                                           \
      hipLaunchKernelGGL((x2d::gemm_halo_kernel<32, E, 1>), grid, dim3(256), 0, s, p, q, M, (int)Ng, tm, tn, nch / 9, xpart, nx); \
    else                                                                                                          \
      hipLaunchKernelGGL((x2d::gemm_halo_kernel<16, E, 1>), grid, dim3(256), 0, s, p, q, M, (int)Ng, tm, tn, nch / 9, xpart, nx); \
  } else {                                                                                                        \
    /* LDS-DMA staging of both operands, one tap x 32 channels per chunk (conv_x2d.h) */                          \
    hipLaunchKernelGGL((x2d::gemm_kernel<TAPS, 128, E>), grid, dim3(256), 0, s, p, q, M, (int)Ng, tm, tn, nch,    \
                       r.chunks_per_split, xpart, nx);                                                            \
  }
#define STK_X2_LAUNCH_E(E)                                                                                        \
  if (planes) { if (p.taps == 9) { STK_PL_LAUNCH(E, 9); } else { STK_PL_LAUNCH(E, 1); } }                         \
  else if (p.taps == 9) { if (S2 > 0) STK_X2_LAUNCH(E, true, 9); else STK_X2_LAUNCH(E, false, 9); }               \
  else { if (S2 > 0) STK_X2_LAUNCH(E, true, 1); else STK_X2_LAUNCH(E, false, 1); }
    if (r.splits == 1) {
      STK_X2_LAUNCH_E(EP)
      STK_CHECK_LAUNCH();
      return STK_OK;
    }
    STK_X2_LAUNCH_E(EpSlab)
    STK_CHECK_LAUNCH();
    launch_slab_sum(p, r.splits, M, Ng, dgrad, s);
#undef STK_X2_LAUNCH_E
#undef STK_X2_LAUNCH
#undef STK_PL_LAUNCH
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
}

// Weight gradient on the split kernel: 3x3 / stride 1 / pad 1 or 1x1 / stride 1, power-of-two maps of >= 8 columns and
// >= 32 pixels, enough channels to fill 128-wide tiles.  One GEMM per tap, K (= pixels) split so that <= 512
// workgroups run.
struct X3WgradPlan { int ok; int splits; int chunks_per_split; long slab; int rows3; };
inline X3WgradPlan x3_wgrad_plan(int C1, int C2, int N, int Cout, int H, int W, int OH, int OW, int KH, int KW, int stride,
                                 int pad) {
  X3WgradPlan q = {0, 0, 0, 0, 0};
  const int Cin = C1 + C2;
  const long K = (long)N * H * W;
  const int cmax = Cout > C1 ? (Cout > C2 ? Cout : C2) : (C1 > C2 ? C1 : C2);
  if (K * cmax * 4 >= 0x7fffffffL || (C2 > 0 && C1 % 32)) return q;    // 32-bit buffer offsets; a wave's rows share a tensor
  const bool pow2 = (W & (W - 1)) == 0 && ((H * W) & (H * W - 1)) == 0;
  const bool geom = (KH == 3 && KW == 3 && pad == 1) || (KH == 1 && KW == 1 && pad == 0);
  if (!geom || stride != 1 || OH != H || OW != W || !pow2 || W < 4 || H * W < 16 || Cin < 64 || Cout < 64 ||
      K > 0x7fffffffL || K % 32)
    return q;
  const int taps = KH * KW;
  // 3x3 on maps of >= 8 columns: three taps (one kernel row) per workgroup on 128 x 64 tiles (wgrad3_kernel)
  q.rows3 = (taps == 9 && W >= 8 && (C2 == 0 || C1 % 16 == 0)) ? 1 : 0;
  const long tiles = q.rows3 ? 3L * stk_cdiv(Cout, 128) * stk_cdiv(Cin, 64)
                             : (long)taps * stk_cdiv(Cout, 128) * stk_cdiv(Cin, 128);
  // a 1x1 layer with few tiles needs many K splits: with ONE wave of 256 workgroups the slab traffic still pays
  // (256->256 NIN, 4 tiles: 54 us against 74 us for the f32-input kernel; 384->128 at 32x32, 3 tiles: 110 against
  // 255; 128->256 at 16x16, 2 tiles: 32 against 41); with 512 workgroups it did not
  if (taps == 1 && tiles < 2) return q;
  const long chunks = K / 32;
  constexpr long w1_wgs = 256;                // (128 / 192 were neutral inside the step, profiles/r05_insitu_sweeps.txt)
  long splits = (taps == 1 && tiles < 6 ? w1_wgs : 512) / tiles;
  if (splits > chunks / 8) splits = chunks / 8;
  if (splits < 1) splits = 1;
  q.chunks_per_split = (int)((chunks + splits - 1) / splits);
  q.splits = (int)((chunks + q.chunks_per_split - 1) / q.chunks_per_split);
  q.slab = (long)Cout * Cin * taps;
  q.ok = 1;
  return q;
}

struct WgradPlan { int big; int splits; int k_per_split; long slab; int mode9; };
// mode9 (all-taps kernel) preconditions; the caller also checks stride 1, pad 1, one source, OH == H, OW == W.
inline int wgrad9_cols(int OH, int OW, int KH, int KW) {
  if (KH != 3 || KW != 3 || ((long)OH * OW) % 32 != 0) return 0;
  if (OW % 32 == 0) return 32;
  if (OW == 16 || OW == 8) return OW;
  return 0;
}
inline WgradPlan wgrad_plan(int Cin, int N, int Cout, int OH, int OW, int KH, int KW, bool allow9 = true) {
  WgradPlan q;
  q.mode9 = allow9 ? wgrad9_cols(OH, OW, KH, KW) : 0;
  if (q.mode9) {
    const long K = (long)N * OH * OW;
    const long tiles = (long)stk_cdiv(Cout, 128) * stk_cdiv(Cin, 32);
    const long chunks = K / 32;
    // one workgroup per CU is resident (144 accumulator + ~120 other registers): aim for at most 256 blocks and
    // never for "one wave of blocks plus a few" (264 blocks on 256 CUs ran 2x slower than 252)
    long splits = 256 / tiles;
    if (splits > chunks / 16) splits = chunks / 16;
    if (splits < 1) splits = 1;
    const long cps = (chunks + splits - 1) / splits;
    q.big = 1;
    q.k_per_split = (int)(cps * 32);
    q.splits = (int)((K + q.k_per_split - 1) / q.k_per_split);
    q.slab = (long)Cout * Cin * 9;
    return q;
  }
  const int taps = KH * KW;
  const long K = (long)N * OH * OW;
  // 128x128 tiles halve the operand traffic per FLOP, but a 1x1 layer has too few of them: with so few tiles
  // the K split would have to be so fine that writing / re-reading the partial slabs dominates.
  const long tiles128 = (long)stk_cdiv(Cout, 128) * stk_cdiv(Cin, 128) * taps;
  q.big = (Cout >= 96 && Cin >= 96 && tiles128 >= 9) ? 1 : 0;
  const int T = q.big ? 128 : 64;
  const long tiles = (long)stk_cdiv(Cout, T) * stk_cdiv(Cin, T) * taps;
  const long chunks = (K + 31) / 32;
  long splits = (512 + tiles - 1) / tiles;
  if (splits > chunks / 16) splits = chunks / 16;      // >= 16 chunks (512 pixels) of work per block
  if (splits < 1) splits = 1;
  const long cps = (chunks + splits - 1) / splits;
  q.k_per_split = (int)(cps * 32);
  q.splits = (int)((K + q.k_per_split - 1) / q.k_per_split);
  q.slab = (long)Cout * Cin * taps;
  return q;
}

}  // namespace

extern "C" {

static int fwd_impl(const float* x1, int C1, const float* x2, int C2, const float* w, int w_layout,
                    const float* bias, const float* temb, int temb_stride, const float* res, float out_div,
                    float* y, int N, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad,
                    const void* wp, float* amax, void* ws, long ws_bytes, void* stream, bool x_rec_valid);

int stk_conv2d_fwd_wp_f32(const float* x1, int C1, const float* x2, int C2, const float* w, int w_layout,
                          const float* bias, const float* temb, int temb_stride, const float* res, float out_div,
                          float* y, int N, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad,
                          const void* wp, float* amax, void* ws, long ws_bytes, void* stream) {
  return fwd_impl(x1, C1, x2, C2, w, w_layout, bias, temb, temb_stride, res, out_div, y, N, H, W, Cout, OH, OW, KH, KW, stride,
                  pad, wp, amax, ws, ws_bytes, stream, false);
}
/* ... with the |x1| / |x2| scale records already in amax[0..256) / amax[256..512) (stk_gn_fwd_pl_max_f32): no |x| pass */
int stk_conv2d_fwd_rec_f32(const float* x1, int C1, const float* x2, int C2, const float* w, int w_layout,
                           const float* bias, const float* temb, int temb_stride, const float* res, float out_div,
                           float* y, int N, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad,
                           const void* wp, float* amax, void* ws, long ws_bytes, void* stream) {
  if (!amax) return STK_EINVAL;
  return fwd_impl(x1, C1, x2, C2, w, w_layout, bias, temb, temb_stride, res, out_div, y, N, H, W, Cout, OH, OW, KH, KW, stride,
                  pad, wp, amax, ws, ws_bytes, stream, true);
}
static int fwd_impl(const float* x1, int C1, const float* x2, int C2, const float* w, int w_layout,
                    const float* bias, const float* temb, int temb_stride, const float* res, float out_div,
                    float* y, int N, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad,
                    const void* wp, float* amax, void* ws, long ws_bytes, void* stream, bool x_rec_valid) {
  if (!x1 || !w || !y || (C2 > 0 && !x2) || out_div == 0.f || (w_layout != 0 && w_layout != 1) ||
      (w_layout == 1 && (KH != 1 || KW != 1)))
    return STK_EINVAL;
  ConvP p = {};
  int rc = fill_common(p, N, H, W, C1, C2, Cout, OH, OW, KH, KW, stride, pad);
  if (rc) return rc;
  p.x1 = x1; p.x2 = C2 > 0 ? x2 : x1; p.w = w; p.w_layout = w_layout; p.bias = bias; p.temb = temb;
  p.temb_stride = temb_stride; p.res = res; p.inv_div = 1.f / out_div; p.use_div = out_div != 1.f; p.y = y;
  const int K = p.Cin * p.taps;
  const long Ng = (long)N * p.OHW;
  const bool big = use_big_tile(Cout, Ng, 1);
  hipStream_t s = (hipStream_t)stream;
  if (thin::geometry_ok(p) && Ng <= 0x7fffffffL) {          // a 3-channel side: streaming kernels (conv_thin.h)
    const unsigned nblk = (unsigned)stk_cdiv(Ng, 256L);
    if (C2 == 0 && p.Cin <= 4) {
      thin::Args a = {x1, w, (long)p.Cin * p.taps, (long)p.taps, 0, p.Cin, Cout, p.taps, 0};
      const dim3 grid(nblk, (unsigned)stk_cdiv(Cout, thin::OB));
      if (p.taps == 9) hipLaunchKernelGGL((thin::thin_in_kernel<9>), grid, dim3(256), 0, s, p, a);
      else hipLaunchKernelGGL((thin::thin_in_kernel<1>), grid, dim3(256), 0, s, p, a);
      STK_CHECK_LAUNCH();
      return STK_OK;
    }
    if (Cout <= 4) {
      const unsigned nblk64 = (unsigned)stk_cdiv(Ng, 64L);
      if (p.taps == 9) hipLaunchKernelGGL((thin::thin_out_kernel<9>), dim3(nblk64), dim3(256), 0, s, p, Cout);
      else hipLaunchKernelGGL((thin::thin_out_kernel<1>), dim3(nblk64), dim3(256), 0, s, p, Cout);
      STK_CHECK_LAUNCH();
      return STK_OK;
    }
  }
  const X3Plan xr = x3_plan(p, p.Cin, C1, C2, Cout, Ng);
  if (ws && xr.ok && ws_bytes >= x3_ws_bytes(xr, Cout, p.Cin, p.taps))
    return launch_x3<EpFwd>(p, xr, x1, C1, x2, C2, Cout, Ng, 0, ws, s, wp, amax, nullptr, nullptr, x_rec_valid);
  if (wp) return STK_EINVAL;      // prepared weights exist only for the shapes stk_conv2d_wp_bytes reports
  if (p.taps == 9) {
    using CB = Cfg<128, 128, 36>; using CS = Cfg<64, 64, 36>;
    if (C2 > 0) {
      if (big) return launch<CB, ConvP, AFwdK<CB>, BFwd<CB, 9, true>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
      return launch<CS, ConvP, AFwdK<CS>, BFwd<CS, 9, true>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
    }
    if (big) return launch<CB, ConvP, AFwdK<CB>, BFwd<CB, 9, false>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
    return launch<CS, ConvP, AFwdK<CS>, BFwd<CS, 9, false>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
  }
  using CB = Cfg<128, 128, 32>; using CS = Cfg<64, 64, 32>;
  // 1x1: B loader always with the dual-source form (the shortcut convs of the up path read a concat)
  if (w_layout == 0) {
    if (K % 8 == 0) {      // row-chunk weight loader needs whole 8-float chunks
      if (big) return launch<CB, ConvP, AFwdK<CB>, BFwd<CB, 1, true>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
      return launch<CS, ConvP, AFwdK<CS>, BFwd<CS, 1, true>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
    }
    return launch<CS, ConvP, AFwdGen<CS>, BFwd<CS, 1, true>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
  }
  if (big) return launch<CB, ConvP, AFwdNin<CB>, BFwd<CB, 1, true>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
  return launch<CS, ConvP, AFwdNin<CS>, BFwd<CS, 1, true>, EpFwd>(p, Cout, Ng, K, K, 1, 1, s);
}

int stk_conv2d_fwd_f32(const float* x1, int C1, const float* x2, int C2, const float* w, int w_layout,
                       const float* bias, const float* temb, int temb_stride, const float* res, float out_div,
                       float* y, int N, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad,
                       void* ws, long ws_bytes, void* stream) {
  return stk_conv2d_fwd_wp_f32(x1, C1, x2, C2, w, w_layout, bias, temb, temb_stride, res, out_div, y, N, H, W, Cout, OH, OW,
                               KH, KW, stride, pad, nullptr, nullptr, ws, ws_bytes, stream);
}

static int dgrad_impl(const float* dy, const float* w, int w_layout, float* dx1, int C1, float beta1, float* dx2,
                      int C2, float beta2, float alpha, int N, int H, int W, int Cout, int OH, int OW, int KH,
                      int KW, int stride, int pad, const void* wp, float* amax, void* ws, long ws_bytes,
                      void* stream, bool dy_rec_valid);
int stk_conv2d_dgrad_wp_f32(const float* dy, const float* w, int w_layout, float* dx1, int C1, float beta1, float* dx2,
                            int C2, float beta2, float alpha, int N, int H, int W, int Cout, int OH, int OW, int KH,
                            int KW, int stride, int pad, const void* wp, float* amax, void* ws, long ws_bytes,
                            void* stream) {
  return dgrad_impl(dy, w, w_layout, dx1, C1, beta1, dx2, C2, beta2, alpha, N, H, W, Cout, OH, OW, KH, KW, stride, pad, wp,
                    amax, ws, ws_bytes, stream, false);
}
/* ... with the |dy| scale record already in amax[512..768) (written by stk_bias_grad_amax*_f32): no |dy| pass */
int stk_conv2d_dgrad_rec_f32(const float* dy, const float* w, int w_layout, float* dx1, int C1, float beta1, float* dx2,
                             int C2, float beta2, float alpha, int N, int H, int W, int Cout, int OH, int OW, int KH,
                             int KW, int stride, int pad, const void* wp, float* amax, void* ws, long ws_bytes,
                             void* stream) {
  if (!amax) return STK_EINVAL;
  return dgrad_impl(dy, w, w_layout, dx1, C1, beta1, dx2, C2, beta2, alpha, N, H, W, Cout, OH, OW, KH, KW, stride, pad, wp,
                    amax, ws, ws_bytes, stream, true);
}
static int dgrad_impl(const float* dy, const float* w, int w_layout, float* dx1, int C1, float beta1, float* dx2,
                      int C2, float beta2, float alpha, int N, int H, int W, int Cout, int OH, int OW, int KH,
                      int KW, int stride, int pad, const void* wp, float* amax, void* ws, long ws_bytes,
                      void* stream, bool dy_rec_valid) {
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
  if (thin::geometry_ok(p) && Cout <= 4 && Ng <= 0x7fffffffL) {     // data gradient of a thin-output layer
    thin::Args a = {dy, w, (long)p.taps, (long)Cin * p.taps, 1, Cout, Cin, p.taps, 1};
    const dim3 grid((unsigned)stk_cdiv(Ng, 256L), (unsigned)stk_cdiv(Cin, thin::OB));
    if (p.taps == 9) hipLaunchKernelGGL((thin::thin_in_kernel<9>), grid, dim3(256), 0, s, p, a);
    else hipLaunchKernelGGL((thin::thin_in_kernel<1>), grid, dim3(256), 0, s, p, a);
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
  const X3Plan xr = x3_plan(p, Cout, Cout, 0, Cin, Ng);
  if (ws && xr.ok && ws_bytes >= x3_ws_bytes(xr, Cin, Cout, p.taps))
    return launch_x3<EpDgrad>(p, xr, dy, Cout, nullptr, 0, Cin, Ng, 1, ws, s, wp, amax, nullptr, nullptr, dy_rec_valid);
  if (wp) return STK_EINVAL;
  if (p.taps == 9) {
    using CB = Cfg<128, 128, 36>; using CS = Cfg<64, 64, 36>;
    if (big) return launch<CB, ConvP, ADgrad9<CB>, BDgrad<CB, 9>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
    return launch<CS, ConvP, ADgrad9<CS>, BDgrad<CS, 9>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
  }
  using CB = Cfg<128, 128, 32>; using CS = Cfg<64, 64, 32>;
  if (w_layout == 0) {       // A(m=ci,k=co) = w[co*Cin + ci]
    if (big) return launch<CB, ConvP, ADgrad1<CB>, BDgrad<CB, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
    return launch<CS, ConvP, ADgrad1<CS>, BDgrad<CS, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
  }
  // NIN: A(m=ci,k=co) = w[ci*Cout + co]
  if (K % 8 != 0) return launch<CS, ConvP, ADgradNinGen<CS>, BDgrad<CS, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
  if (big) return launch<CB, ConvP, ADgradNin<CB>, BDgrad<CB, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
  return launch<CS, ConvP, ADgradNin<CS>, BDgrad<CS, 1>, EpDgrad>(p, Cin, Ng, K, K, 1, 1, s);
}

int stk_conv2d_dgrad_f32(const float* dy, const float* w, int w_layout, float* dx1, int C1, float beta1, float* dx2,
                         int C2, float beta2, float alpha, int N, int H, int W, int Cout, int OH, int OW, int KH,
                         int KW, int stride, int pad, void* ws, long ws_bytes, void* stream) {
  return stk_conv2d_dgrad_wp_f32(dy, w, w_layout, dx1, C1, beta1, dx2, C2, beta2, alpha, N, H, W, Cout, OH, OW, KH, KW, stride,
                                 pad, nullptr, nullptr, ws, ws_bytes, stream);
}

/* ---- planes (conv_pl.h): activations pre-split into two fp16 planes, [split][n][c / 32][pixel][c % 32] ---- */
long stk_planes_bytes(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return 2 * pl::plane_bytes(N, C, HW);
}

int stk_amax_partial_f32(const float* x, long n, float* part, void* stream) {
  if (!x || !part || n <= 0) return STK_EINVAL;
  hipLaunchKernelGGL(x2::amax_partial_kernel, dim3(x2::NPART), dim3(x2::AMAX_THREADS), 0, (hipStream_t)stream, x, n, part);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_split_planes_f32(const float* x, int N, int C, int HW, const float* amax, int namax, void* planes, void* stream) {
  if (!x || !amax || !planes || N <= 0 || C <= 0 || HW <= 0 || namax <= 0) return STK_EINVAL;
  if (2 * pl::plane_bytes(N, C, HW) >= 0x7fffffffL) return STK_EUNSUPPORTED;      // 32-bit buffer offsets in the consumers
  const long blocks = (long)N * ((C + 31) / 32) * stk_cdiv(HW, pl::SP_PIX);
  hipLaunchKernelGGL(pl::split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, N, C, HW, amax,
                     namax, static_cast<unsigned char*>(planes), pl::plane_bytes(N, C, HW));
  STK_CHECK_LAUNCH();
  return STK_OK;
}

/* 1 when the forward (dir 0) / data-gradient (dir 1) call of this shape can read its activation operand (x resp. dy) as
 * planes: exactly the shapes that take the fp16 split kernel with a single-source operand. */
int stk_conv2d_pl_ok(int dir, int C1, int C2, int N, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
  ConvP p = {};
  if (fill_common(p, N, H, W, C1, C2, Cout, H, W, KH, KW, stride, pad)) return 0;
  if (stride != 1 || pad != KH / 2) return 0;
  const long Ng = (long)N * p.HW;
  if (dir == 0) {
    if (C2 > 0 || p.Cin <= 4 || Cout <= 4) return 0;        // two sources / thin-side streaming kernels
    return x3_plan(p, p.Cin, C1, 0, Cout, Ng).ok;
  }
  if (dir == 1) {
    if (Cout <= 4 || p.Cin <= 4) return 0;
    return x3_plan(p, Cout, Cout, 0, p.Cin, Ng).ok;
  }
  return 0;
}

/* number of K splits the plane-operand forward (dir 0) / data-gradient (dir 1) call of this shape runs with: 1 = one
 * launch of x2d::gemm_kernel<.., EpFwd / EpDgrad>, > 1 = EpSlab partial tiles + a slab-sum launch (small maps),
 * 0 = the shape does not take plane operands.  (Profiler labels: bench.py names kernels by their rocprof symbol.) */
int stk_conv2d_pl_ksplit(int dir, int C1, int C2, int N, int H, int W, int Cout, int KH, int KW) {
  if (!stk_conv2d_pl_ok(dir, C1, C2, N, H, W, Cout, KH, KW, 1, KH / 2)) return 0;
  ConvP p = {};
  if (fill_common(p, N, H, W, C1, C2, Cout, H, W, KH, KW, 1, KH / 2)) return 0;
  const long Ng = (long)N * p.HW;
  return dir == 0 ? x3_plan_pl(p, p.Cin, Cout, Ng).splits : x3_plan_pl(p, Cout, p.Cin, Ng).splits;
}

int stk_conv2d_fwd_pl_f32(const void* xpl, const float* xamax, int C, const float* w, int w_layout, const float* bias,
                          const float* temb, int temb_stride, const float* res, float out_div, float* y, int N, int H,
                          int W, int Cout, int KH, int KW, const void* wp, void* ws, long ws_bytes, void* stream) {
  if (!xpl || !xamax || !w || !y || out_div == 0.f || (w_layout != 0 && w_layout != 1) || (w_layout == 1 && KH != 1))
    return STK_EINVAL;
  if (!stk_conv2d_pl_ok(0, C, 0, N, H, W, Cout, KH, KW, 1, KH / 2)) return STK_EUNSUPPORTED;
  ConvP p = {};
  fill_common(p, N, H, W, C, 0, Cout, H, W, KH, KW, 1, KH / 2);
  p.w = w; p.w_layout = w_layout; p.bias = bias; p.temb = temb; p.temb_stride = temb_stride; p.res = res;
  p.inv_div = 1.f / out_div; p.use_div = out_div != 1.f; p.y = y;
  const long Ng = (long)N * p.HW;
  const X3Plan xr = x3_plan_pl(p, C, Cout, Ng);
  if (!ws || ws_bytes < x3_ws_bytes(xr, Cout, C, p.taps)) return STK_EINVAL;
  return launch_x3<EpFwd>(p, xr, nullptr, C, nullptr, 0, Cout, Ng, 0, ws, (hipStream_t)stream, wp, nullptr, xpl, xamax);
}

int stk_conv2d_dgrad_pl_f32(const void* dypl, const float* dyamax, const float* w, int w_layout, float* dx1, int C1,
                            float beta1, float* dx2, int C2, float beta2, float alpha, int N, int H, int W, int Cout,
                            int KH, int KW, const void* wp, void* ws, long ws_bytes, void* stream) {
  if (!dypl || !dyamax || !w || (!dx1 && !dx2) || (w_layout != 0 && w_layout != 1) || (w_layout == 1 && KH != 1))
    return STK_EINVAL;
  if (!stk_conv2d_pl_ok(1, C1, C2, N, H, W, Cout, KH, KW, 1, KH / 2)) return STK_EUNSUPPORTED;
  ConvP p = {};
  fill_common(p, N, H, W, C1, C2, Cout, H, W, KH, KW, 1, KH / 2);
  p.w = w; p.w_layout = w_layout; p.dx1 = dx1; p.dx2 = C2 > 0 ? dx2 : nullptr;
  p.beta1 = beta1; p.beta2 = beta2; p.alpha = alpha;
  const long Ng = (long)N * p.HW;
  const X3Plan xr = x3_plan_pl(p, Cout, p.Cin, Ng);
  if (!ws || ws_bytes < x3_ws_bytes(xr, p.Cin, Cout, p.taps)) return STK_EINVAL;
  return launch_x3<EpDgrad>(p, xr, nullptr, Cout, nullptr, 0, p.Cin, Ng, 1, ws, (hipStream_t)stream, wp, nullptr, dypl, dyamax);
}

/* 3x3 / stride 1 / pad 1 weight gradient with x and dy given as planes (conv_x2w.h) */
int stk_conv2d_wgrad_pl_ok(int N, int H, int W, int Cin, int Cout) { return x2w::plan(N, H, W, Cin, Cout).ok; }

long stk_conv2d_wgrad_pl_ws_bytes(int N, int H, int W, int Cin, int Cout) {
  // (covers every workgroup count a caller may ask for: the slabs of the deepest K split)
  long m = 0;
  for (int wgs : {0, 256, 512, 768, 1024}) {
    const x2w::Plan q = x2w::plan(N, H, W, Cin, Cout, wgs);
    if (q.ok && (long)q.splits * q.slab > m) m = (long)q.splits * q.slab;
  }
  return m ? m * 4 + 256 : 0;
}

int stk_conv2d_wgrad_pl_f32(const void* xpl, const float* xrec, const void* dypl, const float* dyrec, float* dw,
                            float alpha, float* ws, long ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream) {
  return stk_conv2d_wgrad_pl_wgs_f32(xpl, xrec, dypl, dyrec, dw, alpha, ws, ws_bytes, N, H, W, Cin, Cout, 0, stream);
}

/* ... with the number of workgroups its K split fills chosen by the caller (0 = the library's default for a launch that shares the chip
 * with another stream; <= 1024).  Same result up to the summation order of the slabs. */
int stk_conv2d_wgrad_pl_wgs_f32(const void* xpl, const float* xrec, const void* dypl, const float* dyrec, float* dw,
                                float alpha, float* ws, long ws_bytes, int N, int H, int W, int Cin, int Cout, int wgs, void* stream) {
  if (!xpl || !xrec || !dypl || !dyrec || !dw || !ws || wgs < 0 || wgs > 1024) return STK_EINVAL;
  const x2w::Plan q = x2w::plan(N, H, W, Cin, Cout, wgs);
  if (!q.ok) return STK_EUNSUPPORTED;
  if (ws_bytes < (long)q.splits * q.slab * 4) return STK_EINVAL;
  x2w::Args a = {};
  a.dypl = static_cast<const unsigned char*>(dypl); a.dyrec = dyrec; a.dy_ps = pl::plane_bytes(N, Cout, H * W);
  a.xpl = static_cast<const unsigned char*>(xpl); a.xrec = xrec; a.x_ps = pl::plane_bytes(N, Cin, H * W);
  if (2 * a.dy_ps >= 0x7fffffffL || 2 * a.x_ps >= 0x7fffffffL) return STK_EUNSUPPORTED;
  a.part = ws; a.part_stride = q.slab;
  a.N = N; a.H = H; a.W = W; a.HW = H * W; a.Cin = Cin; a.Cout = Cout; a.Cob = Cout / 32; a.Cib = Cin / 32;
  a.tiles_co = stk_cdiv(Cout, 128); a.tiles_ci = Cin / 32;
  a.nchunks_total = (int)((long)N * H * W / 32); a.chunks_per_split = q.chunks_per_split;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)(a.tiles_co * a.tiles_ci * q.splits));
  if (q.groups == 2) {
    if (W >= 32) hipLaunchKernelGGL((x2w::wgrad_kernel<32, 2>), grid, dim3(512), 0, s, a);
    else if (W == 16) hipLaunchKernelGGL((x2w::wgrad_kernel<16, 2>), grid, dim3(512), 0, s, a);
    else if (W == 8) hipLaunchKernelGGL((x2w::wgrad_kernel<8, 2>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((x2w::wgrad_kernel<4, 2>), grid, dim3(512), 0, s, a);
  } else if (W >= 32) hipLaunchKernelGGL((x2w::wgrad_kernel<32>), grid, dim3(256), 0, s, a);
  else if (W == 16) hipLaunchKernelGGL((x2w::wgrad_kernel<16>), grid, dim3(256), 0, s, a);
  else if (W == 8) hipLaunchKernelGGL((x2w::wgrad_kernel<8>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((x2w::wgrad_kernel<4>), grid, dim3(256), 0, s, a);
  STK_CHECK_LAUNCH();
  // measured (tools/bench_x2d.py, kernel + reduce, us, old -> new): 32 slabs 118.5 -> 117.0 / 55.1 -> 51.1, 64: 74.4 -> 72.3, 16: 206.0 ->
  // 201.8, 8: 44.8 -> 39.7, but 128 slabs 124.1 -> 130.8 (1152 four-byte loads per thread): the coalesced form up to 64 slabs
  if (q.splits <= 64)
    hipLaunchKernelGGL(splitk_reduce9_kernel, dim3((unsigned)stk_cdiv((long)Cout * Cin, 256L)), dim3(256), 0, s, ws, dw, (long)Cout * Cin,
                       q.splits, q.slab, alpha);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(stk_ew_grid((q.slab + 3) / 4)), dim3(256), 0, s, ws, dw, q.slab, q.splits,
                       q.slab, alpha, 0, Cout, Cin, 9);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

/* map width of the halo-tile GEMM (x2d::gemm_halo_kernel<W, ..>) the plane-operand forward / data-gradient call of this
 * shape runs on, 0 = x2d::gemm_kernel (diagnostic: one profiler label per kernel symbol) */
int stk_conv2d_pl_halo(int dir, int C1, int C2, int N, int H, int W, int Cout, int KH, int KW) {
  const int ks = stk_conv2d_pl_ksplit(dir, C1, C2, N, H, W, Cout, KH, KW);
  if (ks <= 0) return 0;
  ConvP p = {};
  if (fill_common(p, N, H, W, C1, C2, Cout, H, W, KH, KW, 1, KH / 2)) return 0;
  return x2d::halo_ok(p, p.taps, ks) ? x2d::halo_cols(W) : 0;
}

/* which kernel family a call with full scratch takes: 0/1 = f32-input MFMA with 64/128 tiles, 2 = bf16 three-way
 * split, 3 = f32-input all-taps weight gradient, 4 = thin-side streaming kernels, 5 = fp16 two-way split.  dir: 0 fwd, 1 dgrad, 2 wgrad. */
int stk_conv2d_variant(int dir, int C1, int C2, int N, int H, int W, int Cout, int OH, int OW, int KH, int KW,
                       int stride, int pad, int w_layout) {
  ConvP p = {};
  if (fill_common(p, N, H, W, C1, C2, Cout, OH, OW, KH, KW, stride, pad)) return -1;
  const int Cin = C1 + C2;
  p.w_layout = w_layout;
  if (dir == 0) {
    const long Ng = (long)N * p.OHW;
    if (thin::geometry_ok(p) && ((C2 == 0 && Cin <= 4) || Cout <= 4)) return 4;
    if (x3_plan(p, Cin, C1, C2, Cout, Ng).ok) return 5;
    return use_big_tile(Cout, Ng, 1) && !(p.taps == 1 && w_layout == 0 && (Cin % 8)) ? 1 : 0;
  }
  if (dir == 1) {
    const long Ng = (long)N * p.HW;
    if (thin::geometry_ok(p) && Cout <= 4) return 4;
    if (x3_plan(p, Cout, Cout, 0, Cin, Ng).ok) return 5;
    return use_big_tile(Cin, Ng, 1) && !(p.taps == 1 && w_layout == 1 && (Cout % 8)) ? 1 : 0;
  }
  if (thin::geometry_ok(p) && ((C2 == 0 && Cin <= 4) || Cout <= 4)) return 4;
  {
    const X3WgradPlan xq = x3_wgrad_plan(C1, C2, N, Cout, H, W, OH, OW, KH, KW, stride, pad);
    if (xq.ok) return 5;
  }
  const bool can9 = stride == 1 && pad == 1 && C2 == 0 && OH == H && OW == W && w_layout == 0;
  const WgradPlan q = wgrad_plan(Cin, N, Cout, OH, OW, KH, KW, can9);
  return q.mode9 ? 3 : q.big;
}

/* Prepared weights (the bf16 three-way split in the A-tile layout of conv_x3.h), so that a caller whose weights change
 * once per optimizer step -- or never, in a sampling loop -- prepares them once instead of once per call.
 * dir: 0 forward, 1 data gradient.  Bytes are 0 for shapes whose call does not take the split kernel. */
long stk_conv2d_wp_bytes(int dir, int C1, int C2, int N, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
  if (dir != 0 && dir != 1) return 0;
  const int v = stk_conv2d_variant(dir, C1, C2, N, H, W, Cout, H, W, KH, KW, stride, pad, 0);
  if (v != 2 && v != 5) return 0;
  const int Cin = C1 + C2;
  return (dir == 0 ? x3::wp_bytes(Cout, Cin, KH * KW) : x3::wp_bytes(Cin, Cout, KH * KW)) + 256;
}

/* Host-side fill of one descriptor; returns the number of (row, k) work items of this layer (the caller passes the
 * maximum over its table to stk_conv2d_wprep_batch) or a negative error.  wp must be 256-byte aligned. */
long stk_conv2d_wp_desc(int dir, const float* w, int w_layout, int Cin, int Cout, int KH, int KW, void* wp,
                        StkWprepDesc* out) {
  if (!w || !wp || !out || (dir != 0 && dir != 1) || (w_layout != 0 && w_layout != 1) || Cin <= 0 || Cout <= 0 ||
      (KH * KW != 9 && KH * KW != 1) || (w_layout == 1 && KH * KW != 1) || ((uintptr_t)wp & 255))
    return STK_EINVAL;
  ConvP p = {};
  p.Cin = Cin; p.Cout = Cout; p.taps = KH * KW; p.w_layout = w_layout;
  long sm, sk;
  x3_weight_strides(p, dir, sm, sk);
  out->w = w; out->wp = wp; out->sm = sm; out->sk = sk;
  out->M = dir ? Cin : Cout; out->Kc = dir ? Cout : Cin; out->Mpad = x3::pad128(out->M); out->taps = p.taps;
  out->flip = dir; out->reserved = 0;
  return (long)out->Mpad * out->Kc;
}

int stk_conv2d_wprep_batch(const StkWprepDesc* descs_dev, int n, long max_items, void* stream) {
  static_assert(sizeof(StkWprepDesc) == sizeof(x2::WprepDesc), "descriptor layout");
  if (!descs_dev || n <= 0 || max_items <= 0 || n > 65535) return STK_EINVAL;
  const dim3 grid((unsigned)stk_cdiv(max_items, 256L), (unsigned)n);
  const x2::WprepDesc* d = reinterpret_cast<const x2::WprepDesc*>(descs_dev);
  hipLaunchKernelGGL(x2::wamax_kernel, dim3(x2::WPART, (unsigned)n), dim3(256), 0, (hipStream_t)stream, d, x2::WprepDesc{});
  hipLaunchKernelGGL(x2::wprep_kernel, grid, dim3(256), 0, (hipStream_t)stream, d, x2::WprepDesc{});
  STK_CHECK_LAUNCH();
  return STK_OK;
}

long stk_conv2d_fwd_ws_bytes(int C1, int C2, int N, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
  ConvP p = {};
  if (fill_common(p, N, H, W, C1, C2, Cout, H, W, KH, KW, stride, pad)) return 0;
  const long a = x3_ws_bytes(x3_plan(p, C1 + C2, C1, C2, Cout, (long)N * H * W), Cout, C1 + C2, p.taps);
  const long b = C2 == 0 ? x3_ws_bytes(x3_plan_pl(p, C1, Cout, (long)N * H * W), Cout, C1, p.taps) : 0;      // the plane-operand plan
  return a > b ? a : b;
}

long stk_conv2d_dgrad_ws_bytes(int C1, int C2, int N, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
  ConvP p = {};
  if (fill_common(p, N, H, W, C1, C2, Cout, H, W, KH, KW, stride, pad)) return 0;
  const long a = x3_ws_bytes(x3_plan(p, Cout, Cout, 0, C1 + C2, (long)N * H * W), C1 + C2, Cout, p.taps);
  const long b = x3_ws_bytes(x3_plan_pl(p, Cout, C1 + C2, (long)N * H * W), C1 + C2, Cout, p.taps);
  return a > b ? a : b;
}

long stk_conv2d_wgrad_ws_bytes(int C1, int C2, int N, int Cout, int OH, int OW, int KH, int KW) {
  const WgradPlan a = wgrad_plan(C1 + C2, N, Cout, OH, OW, KH, KW, true);
  const WgradPlan b = wgrad_plan(C1 + C2, N, Cout, OH, OW, KH, KW, false);
  const X3WgradPlan x = x3_wgrad_plan(C1, C2, N, Cout, OH, OW, OH, OW, KH, KW, 1, KH == 3 ? 1 : 0);
  const long na = (long)a.splits * a.slab, nb = (long)b.splits * b.slab, nx = x.ok ? (long)x.splits * x.slab : 0;
  long m = na > nb ? na : nb;
  if (C2 == 0 && C1 <= 4) { const long t = (long)thin::wgrad_slabs(N, (long)OH * OW, Cout) * Cout * C1 * KH * KW; m = t > m ? t : m; }
  if (Cout <= 4) { const long t = (long)thin::wgrad_slabs(N, (long)OH * OW, C1 + C2) * Cout * (C1 + C2) * KH * KW; m = t > m ? t : m; }
  return (m > nx ? m : nx) * 4 + 256 + 256 + 3L * x2::NPART * 4;      // + partial maxima of dy, x1, x2
}

int stk_conv2d_wgrad_amax_f32(const float* x1, int C1, const float* x2, int C2, const float* dy, float* dw, int w_layout,
                              float alpha, float* ws, long ws_bytes, int N, int H, int W, int Cout, int OH, int OW, int KH,
                              int KW, int stride, int pad, const float* amax, int have, void* stream) {
  if (!x1 || !dy || !dw || !ws || (C2 > 0 && !x2) || (w_layout != 0 && w_layout != 1) ||
      (w_layout == 1 && (KH != 1 || KW != 1)))
    return STK_EINVAL;
  ConvP p = {};
  int rc = fill_common(p, N, H, W, C1, C2, Cout, OH, OW, KH, KW, stride, pad);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  p.w_layout = w_layout;
  if (thin::geometry_ok(p) && ((C2 == 0 && p.Cin <= 4) || Cout <= 4)) {      // a 3-channel side (conv_thin.h)
    const bool stem = C2 == 0 && p.Cin <= 4;
    thin::WArgs a = {};
    if (stem) { a.thin = x1; a.b1 = dy; a.b2 = dy; a.B1 = Cout; a.B2 = 0; a.CT = p.Cin; a.head = 0; }
    else { a.thin = dy; a.b1 = x1; a.b2 = C2 > 0 ? x2 : x1; a.B1 = C1; a.B2 = C2; a.CT = Cout; a.head = 1; }
    const int BC = a.B1 + a.B2, S = thin::wgrad_slabs(N, p.HW, BC);
    a.pxb = (int)((p.HW + 255) / 256); a.items = N * a.pxb; a.part = ws; a.slab = (long)Cout * p.Cin * p.taps;
    if (ws_bytes < (long)S * a.slab * 4) return STK_EINVAL;
    const dim3 grid((unsigned)S, (unsigned)stk_cdiv(BC, 4));
    if (p.taps == 9) {
      if (a.CT == 3) hipLaunchKernelGGL((thin::thin_wgrad_kernel<9, 3>), grid, dim3(256), 0, s, p, a);
      else hipLaunchKernelGGL((thin::thin_wgrad_kernel<9, 4>), grid, dim3(256), 0, s, p, a);
    } else {
      hipLaunchKernelGGL((thin::thin_wgrad_kernel<1, 4>), grid, dim3(256), 0, s, p, a);
    }
    STK_CHECK_LAUNCH();
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(stk_ew_grid((a.slab + 3) / 4)), dim3(256), 0, s, ws, dw, a.slab, S, a.slab,
                       alpha, 0, Cout, p.Cin, p.taps);
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
  const X3WgradPlan xq = x3_wgrad_plan(C1, C2, N, Cout, H, W, OH, OW, KH, KW, stride, pad);
  if (xq.ok && ws_bytes >= (long)xq.splits * xq.slab * 4 + 256 + 3L * x2::NPART * 4) {      // slabs + the |dy| / |x| partial maxima
    p.x1 = x1; p.x2 = C2 > 0 ? x2 : x1; p.dy = dy; p.w_layout = w_layout; p.part = ws; p.part_stride = xq.slab;
    const int tm = stk_cdiv(Cout, 128), tn = stk_cdiv(p.Cin, 128);
    const int nch = (int)((long)N * p.HW / 32);
    if (xq.rows3) {
      const int tn64 = stk_cdiv(p.Cin, 64);
      const dim3 grid3((unsigned)(3 * tm * tn64 * xq.splits));
      float* parts = reinterpret_cast<float*>(((uintptr_t)(ws + (long)xq.splits * xq.slab) + 255) & ~(uintptr_t)255);
      {
        // fp16 two-way split of both operands (conv_x2.h): |dy| and |x| maxima first
        const dim3 ab(x2::NPART), at(x2::AMAX_THREADS);
        // maxima the layer's forward (x) / data-gradient (dy) calls left in `amax` are reused, the others taken here
        const float* dyp = parts;
        const float* xp = parts + x2::NPART;
        if (amax && (have & 2)) dyp = amax + 2 * x2::NPART;
        else hipLaunchKernelGGL(x2::amax_partial_kernel, ab, at, 0, s, dy, (long)N * Cout * p.OHW, parts);
        if (amax && (have & 1)) {
          xp = amax;
        } else {
          hipLaunchKernelGGL(x2::amax_partial_kernel, ab, at, 0, s, x1, (long)N * C1 * p.HW, parts + x2::NPART);
          if (C2 > 0) hipLaunchKernelGGL(x2::amax_partial_kernel, ab, at, 0, s, x2, (long)N * C2 * p.HW, parts + 2 * x2::NPART);
        }
        const int nx = C2 > 0 ? 2 * x2::NPART : x2::NPART;
        if (C2 > 0) hipLaunchKernelGGL((x2::wgrad3_kernel<true>), grid3, dim3(256), 0, s, p, tm, tn64, nch, xq.chunks_per_split,
                                       dyp, xp, nx);
        else hipLaunchKernelGGL((x2::wgrad3_kernel<false>), grid3, dim3(256), 0, s, p, tm, tn64, nch, xq.chunks_per_split,
                                dyp, xp, nx);
      }
      STK_CHECK_LAUNCH();
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(stk_ew_grid((xq.slab + 3) / 4)), dim3(256), 0, s, ws, dw, xq.slab, xq.splits,
                         xq.slab, alpha, w_layout, Cout, p.Cin, p.taps);
      STK_CHECK_LAUNCH();
      return STK_OK;
    }
    const dim3 grid((unsigned)(p.taps * tm * tn * xq.splits));
    {
      // fp16 two-way split of both operands (x2::wgemm_kernel): maxima as for the three-taps kernel above
      float* parts = reinterpret_cast<float*>(((uintptr_t)(ws + (long)xq.splits * xq.slab) + 255) & ~(uintptr_t)255);
      const dim3 ab(x2::NPART), at(x2::AMAX_THREADS);
      const float* dyp = parts;
      const float* xp = parts + x2::NPART;
      if (amax && (have & 2)) dyp = amax + 2 * x2::NPART;
      else hipLaunchKernelGGL(x2::amax_partial_kernel, ab, at, 0, s, dy, (long)N * Cout * p.OHW, parts);
      if (amax && (have & 1)) {
        xp = amax;
      } else {
        hipLaunchKernelGGL(x2::amax_partial_kernel, ab, at, 0, s, x1, (long)N * C1 * p.HW, parts + x2::NPART);
        if (C2 > 0) hipLaunchKernelGGL(x2::amax_partial_kernel, ab, at, 0, s, x2, (long)N * C2 * p.HW, parts + 2 * x2::NPART);
      }
      const int nx = C2 > 0 ? 2 * x2::NPART : x2::NPART;
#define STK_X2_WGRAD1(BLOADER)                                                                                              \
  hipLaunchKernelGGL((x2::wgemm_kernel<x2::RowsU<false, false>, BLOADER, EpWgrad, true>), grid, dim3(256), 0, s, p, Cout,  \
                     p.Cin, tm, tn, nch, xq.chunks_per_split, p.taps, dyp, xp, nx)
#define STK_COMMA ,
      if (p.taps == 1) { if (C2 > 0) STK_X2_WGRAD1(x2::RowsU<true STK_COMMA true>); else STK_X2_WGRAD1(x2::RowsU<true STK_COMMA false>); }
      else if (W >= 16) { if (C2 > 0) STK_X2_WGRAD1(x2::RowsB<true STK_COMMA 16>); else STK_X2_WGRAD1(x2::RowsB<false STK_COMMA 16>); }
      else if (W == 8) { if (C2 > 0) STK_X2_WGRAD1(x2::RowsB<true STK_COMMA 8>); else STK_X2_WGRAD1(x2::RowsB<false STK_COMMA 8>); }
      else { if (C2 > 0) STK_X2_WGRAD1(x2::RowsB<true STK_COMMA 4>); else STK_X2_WGRAD1(x2::RowsB<false STK_COMMA 4>); }
#undef STK_COMMA
#undef STK_X2_WGRAD1
      STK_CHECK_LAUNCH();
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(stk_ew_grid((xq.slab + 3) / 4)), dim3(256), 0, s, ws, dw, xq.slab, xq.splits,
                         xq.slab, alpha, w_layout, Cout, p.Cin, p.taps);
      STK_CHECK_LAUNCH();
      return STK_OK;
    }
  }
  const bool can9 = stride == 1 && pad == 1 && C2 == 0 && OH == H && OW == W && w_layout == 0;
  const WgradPlan q = wgrad_plan(p.Cin, N, Cout, OH, OW, KH, KW, can9);
  if (ws_bytes < (long)q.splits * q.slab * 4) return STK_EINVAL;
  p.x1 = x1; p.x2 = C2 > 0 ? x2 : x1; p.dy = dy; p.w_layout = w_layout; p.part = ws; p.part_stride = q.slab;
  const long Kl = (long)N * p.OHW;
  if (Kl > 0x7fffffffL) return STK_EUNSUPPORTED;
  const int K = (int)Kl;
  using CB = Cfg<128, 128, 32>; using CS = Cfg<64, 64, 32>;
  if (q.mode9) {
    const int tm = stk_cdiv(Cout, 128), tn = stk_cdiv(p.Cin, 32);
    const dim3 grid((unsigned)(tm * tn * q.splits));
    if (q.mode9 == 32) hipLaunchKernelGGL((wgrad9_kernel<32>), grid, dim3(256), 0, s, p, tm, tn, q.k_per_split);
    else if (q.mode9 == 16) hipLaunchKernelGGL((wgrad9_kernel<16>), grid, dim3(256), 0, s, p, tm, tn, q.k_per_split);
    else hipLaunchKernelGGL((wgrad9_kernel<8>), grid, dim3(256), 0, s, p, tm, tn, q.k_per_split);
    STK_CHECK_LAUNCH();
    rc = STK_OK;
  } else if (C2 > 0) {
    if (q.big) rc = launch<CB, ConvP, AWgrad<CB>, BWgrad<CB, true>, EpWgrad>(p, Cout, p.Cin, K, q.k_per_split, q.splits, p.taps, s, true);
    else rc = launch<CS, ConvP, AWgrad<CS>, BWgrad<CS, true>, EpWgrad>(p, Cout, p.Cin, K, q.k_per_split, q.splits, p.taps, s, true);
  } else {
    if (q.big) rc = launch<CB, ConvP, AWgrad<CB>, BWgrad<CB, false>, EpWgrad>(p, Cout, p.Cin, K, q.k_per_split, q.splits, p.taps, s, true);
    else rc = launch<CS, ConvP, AWgrad<CS>, BWgrad<CS, false>, EpWgrad>(p, Cout, p.Cin, K, q.k_per_split, q.splits, p.taps, s, true);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(stk_ew_grid((q.slab + 3) / 4)), dim3(256), 0, s, ws, dw, q.slab, q.splits,
                     q.slab, alpha, w_layout, Cout, p.Cin, p.taps);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_conv2d_wgrad_f32(const float* x1, int C1, const float* x2, int C2, const float* dy, float* dw, int w_layout,
                         float alpha, float* ws, long ws_bytes, int N, int H, int W, int Cout, int OH, int OW, int KH,
                         int KW, int stride, int pad, void* stream) {
  return stk_conv2d_wgrad_amax_f32(x1, C1, x2, C2, dy, dw, w_layout, alpha, ws, ws_bytes, N, H, W, Cout, OH, OW, KH, KW, stride,
                                   pad, nullptr, 0, stream);
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
  // row-chunk loaders need unit k stride and whole 8-float chunks; anything else goes lanes-along-m/n
  const bool ak = sak == 1 && (K % 8) == 0;
  const bool bk = sbk == 1 && sbn != 1 && (K % 8) == 0;
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
