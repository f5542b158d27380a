// groupnorm.hip -- GroupNorm (+SiLU) (+dropout), forward and backward, for gfx950.
//
// Replaces the nn.GroupNorm -> nn.SiLU -> nn.Dropout chains of the reference
// (models/layerspp.py:232,244-245,256,277-278; AttnBlockpp :80,90; models/ncsnpp.py:378-422) with one
// kernel per direction.  The op is HBM-bound: in NCHW one (sample, group) is a contiguous run of
// (C/G)*H*W floats, so a workgroup owns one (sample, group) and streams it with float4 lanes.
//
//  * The input may be the channel-concat of two tensors (the skip connections of the up path,
//    models/ncsnpp.py:368); the concat is never materialised: a group is at most two contiguous
//    segments, one in each source.
//  * Statistics use shifted sums (shift = first element of the group) so that a single pass gives
//    mean and variance without the cancellation of E[x^2]-E[x]^2; wave64 shuffles reduce inside a
//    wave, LDS across the 4 waves.
//  * The normalisation pass re-reads the group right after the statistics pass; a group is
//    16 KB - 1 MB, so the second read is served by the XCD's 4 MB L2 / the 256 MB Infinity Cache
//    rather than HBM (algorithmic traffic: one read + one write of the tensor).
//  * Dropout is a counter-based mask, a pure function of (seed, flat element index) (stk_rng.h), so the
//    backward pass regenerates it instead of storing it.
//  * backward: per (sample, group) workgroup computes the per-channel partial sums of d(gamma),
//    d(beta) into a [N,C,2] scratch and dx; a second tiny kernel folds the scratch over N.  Groups of up to 16384
//    elements (every group of the score network) take the flat register-resident variant.
#include "common.h"

namespace {

struct GnArgs {
  const float* x1; const float* x2;
  int C1, C2;
  const float* gamma; const float* beta;
  int N, HW, G, cpg;
  int act; float drop_p; float keep_scale; unsigned drop_thr;     // drop_thr = stk_drop_threshold(drop_p)
  unsigned long long seed; const unsigned long long* seed_dev;
};

// optional by-products of the flat backward kernel (stk_gn_bwd_out_f32); all NULL = none
struct GnBwdOut {
  float* sum; float* temb; int temb_stride; float scale; float* amax;     // by-products of the final dx1 values
  const float* add; float add_scale;                                      // dx1 += add_scale * add  ([N, C1, HW])
};

// Segment decomposition of group (n, g): channels [c0, c1) -> part in x1, part in x2.
struct Seg {
  const float* p; int len; int c_first;   // len in floats, first channel index (global)
};
__device__ __forceinline__ void group_segments(const GnArgs& a, int n, int g, Seg (&s)[2]) {
  const int c0 = g * a.cpg, c1 = c0 + a.cpg;
  const int e1 = min(c1, a.C1);
  if (c0 < a.C1) {
    s[0].p = a.x1 + ((long)n * a.C1 + c0) * a.HW;
    s[0].len = (e1 - c0) * a.HW;
    s[0].c_first = c0;
  } else {
    s[0].p = nullptr; s[0].len = 0; s[0].c_first = c0;
  }
  const int b2 = max(c0, a.C1);
  if (c1 > a.C1) {
    s[1].p = a.x2 + ((long)n * a.C2 + (b2 - a.C1)) * a.HW;
    s[1].len = (c1 - b2) * a.HW;
    s[1].c_first = b2;
  } else {
    s[1].p = nullptr; s[1].len = 0; s[1].c_first = c1;
  }
}

// u * sigmoid(u) = u / (1 + exp(-u)), to the accuracy of the libm expf + IEEE division it replaces (<= ~1.5 ulp in the
// exponential, a correctly rounded quotient in all but rare cases) at about half their VALU cost (14 instead of ~25
// instructions per element -- the forward GroupNorm + SiLU kernels are VALU-limited, not HBM-limited, at that cost):
//   exp(-u) = 2^t (1 + r ln 2),  t = fl(-u log2 e),  r = the exact residual of that product plus the low part of log2 e
//             (two FMAs), 2^t from v_exp_f32 (1 ulp);
//   u / d   = one v_rcp_f32, one Newton step, and a residual correction of the quotient (FMAs).
// The raw hardware forms had been tried and rejected: __expf without the residual (argument scaling costs |u| 2^-24)
// moved the likelihood ODE's latent to 3e-4 from the reference's fixture, and the bare 1-ulp v_rcp_f32 raised the noise
// floor of the adaptive solver enough for 10 % more function evaluations.  STK_SILU_LIBM (compile time) restores libm.
// Dropout: a float4 item is exactly one quad of the counter RNG (stk_rng.h): one 64-bit mix per item, a 16-bit field
// per element (its flat index is a multiple of 4 because H*W is).
__device__ __forceinline__ float silu_f(float u) {
#ifdef STK_SILU_LIBM
  return u / (1.f + expf(-u));
#else
  const float NL2E_HI = -1.44269502162933349609375f, NL2E_LO = -1.925963033500011e-8f;     // -log2(e) = HI + LO
  const float t = u * NL2E_HI;
  float r = __fmaf_rn(u, NL2E_HI, -t);
  r = __fmaf_rn(u, NL2E_LO, r);
  const float p = __builtin_amdgcn_exp2f(t);
  const float e = fminf(__fmaf_rn(p, r * 0.693147182464599609375f, p), 3.0e38f);          // finite: 1 + e stays finite
  const float d = 1.f + e;
  float rc = __builtin_amdgcn_rcpf(d);
  rc = __fmaf_rn(rc, __fmaf_rn(-d, rc, 1.f), rc);
  const float q = u * rc;
  return __fmaf_rn(__fmaf_rn(-q, d, u), rc, q);
#endif
}

// sigmoid(u) = 1 / (1 + exp(-u)) with the same exponential and the same refined reciprocal as silu_f (backward pass:
// the libm expf + IEEE division cost ~25 VALU instructions per element of a kernel that has ~20 others).
__device__ __forceinline__ float sigmoid_f(float u) {
#ifdef STK_SILU_LIBM
  return 1.f / (1.f + expf(-u));
#else
  const float NL2E_HI = -1.44269502162933349609375f, NL2E_LO = -1.925963033500011e-8f;
  const float t = u * NL2E_HI;
  float r = __fmaf_rn(u, NL2E_HI, -t);
  r = __fmaf_rn(u, NL2E_LO, r);
  const float p = __builtin_amdgcn_exp2f(t);
  const float e = fminf(__fmaf_rn(p, r * 0.693147182464599609375f, p), 3.0e38f);
  const float d = 1.f + e;
  float rc = __builtin_amdgcn_rcpf(d);
  return __fmaf_rn(rc, __fmaf_rn(-d, rc, 1.f), rc);     // one Newton step: <= 1 ulp of 1 / d
#endif
}

// The activation behind the normalisation, by the `act` code of include/stk.h (layers.get_act, models/layers.py:29-41).  SiLU -- the
// only one the shipped configs use -- keeps its hand-tuned form and its place first in the (launch-uniform) branch.
__device__ __forceinline__ float act_f(int act, float u) {
  if (act == STK_ACT_SILU) return silu_f(u);
  if (act == STK_ACT_RELU) return u > 0.f ? u : 0.f;
  if (act == STK_ACT_LRELU) return u > 0.f ? u : 0.2f * u;
  if (act == STK_ACT_ELU) return u > 0.f ? u : expm1f(u);
  return u;
}
// d act(u) / du
__device__ __forceinline__ float act_slope_f(int act, float u) {
  if (act == STK_ACT_SILU) { const float sg = sigmoid_f(u); return sg * (1.f + u * (1.f - sg)); }
  if (act == STK_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (act == STK_ACT_LRELU) return u > 0.f ? 1.f : 0.2f;
  if (act == STK_ACT_ELU) return u > 0.f ? 1.f : expf(u);
  return 1.f;
}

// ---- forward ------------------------------------------------------------------------------------
// Register-resident forward for groups of up to 16384 elements (every group of the 32x32 / 64x64 networks): a thread
// keeps its <= 4 float4 of the group, so the group is read ONCE (the looping kernel below reads it for the statistics
// and again for the normalisation; measured 1.6x the tensor at the memory side).  Same arithmetic as gn_fwd_kernel.
template <int IPT>
__global__ __launch_bounds__(1024) void gn_fwd_flat_kernel(GnArgs a, float* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, float eps) {
  __shared__ float red[32];
  const int ng = blockIdx.x;
  const int n = ng / a.G, g = ng - n * a.G;
  Seg seg[2];
  group_segments(a, n, g, seg);
  const int L = a.cpg * a.HW, L4 = L >> 2, n0 = seg[0].len >> 2;
  const float shift = seg[0].len ? seg[0].p[0] : seg[1].p[0];
  const float4* p0 = reinterpret_cast<const float4*>(seg[0].p);
  const float4* p1 = reinterpret_cast<const float4*>(seg[1].p);
  float4 v[IPT];
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int idx = threadIdx.x + k * blockDim.x;
    v[k] = make_float4(shift, shift, shift, shift);
    if (idx < L4) v[k] = idx < n0 ? p0[idx] : p1[idx - n0];
    const float d0 = v[k].x - shift, d1 = v[k].y - shift, d2 = v[k].z - shift, d3 = v[k].w - shift;
    s[0] += (d0 + d1) + (d2 + d3);
    s[1] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  block_sum<2>(s, red);
  const float inv_l = 1.f / (float)L;
  const float md = s[0] * inv_l;                       // mean - shift
  const float var = fmaxf(s[1] * inv_l - md * md, 0.f);
  const float mean = shift + md;
  const float rstd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[ng] = mean;
    rstd_out[ng] = rstd;
  }
  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;
  const int C = a.C1 + a.C2;
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int idx = threadIdx.x + k * blockDim.x;
    if (idx >= L4) continue;
    const int q = idx < n0 ? 0 : 1, i = idx < n0 ? idx : idx - n0;
    const int c_first = q ? seg[1].c_first : seg[0].c_first;
    float4* o4 = reinterpret_cast<float4*>(y + ((long)n * C + c_first) * a.HW);
    const unsigned long long flat0 = ((unsigned long long)n * C + c_first) * a.HW;
    const int c = c_first + (i * 4) / a.HW;
    const float ga = a.gamma[c], be = a.beta[c];
    float r[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float u = ga * ((r[j] - mean) * rstd) + be;
      float t = a.act ? act_f(a.act, u) : u;
      if (a.drop_p > 0.f) t = stk_drop_field(stk_mix64(seed, (flat0 + (unsigned long long)(i * 4)) >> 2), j) >= a.drop_thr ? t * a.keep_scale : 0.f;
      r[j] = t;
    }
    o4[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// VEC: 4 when HW % 4 == 0 and all pointers are 16-B aligned, else 1.
template <int VEC>
__global__ __launch_bounds__(256) void gn_fwd_kernel(GnArgs a, float* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, float eps) {
  __shared__ float red[16];
  const int ng = blockIdx.x;
  const int n = ng / a.G, g = ng - n * a.G;
  Seg seg[2];
  group_segments(a, n, g, seg);
  const int L = a.cpg * a.HW;
  const float shift = seg[0].len ? seg[0].p[0] : seg[1].p[0];

  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float* p = seg[q].p;
    const int len = seg[q].len;
    if (VEC == 4) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      for (int i = threadIdx.x; i < (len >> 2); i += 256) {
        const float4 v = p4[i];
        const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
        s[0] += (d0 + d1) + (d2 + d3);
        s[1] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    } else {
      for (int i = threadIdx.x; i < len; i += 256) {
        const float d = p[i] - shift;
        s[0] += d;
        s[1] += d * d;
      }
    }
  }
  block_sum<2>(s, red);
  const float inv_l = 1.f / (float)L;
  const float md = s[0] * inv_l;                       // mean - shift
  const float var = fmaxf(s[1] * inv_l - md * md, 0.f);
  const float mean = shift + md;
  const float rstd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[ng] = mean;
    rstd_out[ng] = rstd;
  }

  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;
  const int C = a.C1 + a.C2;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float* p = seg[q].p;
    const int len = seg[q].len;
    if (!len) continue;
    float* o = y + ((long)n * C + seg[q].c_first) * a.HW;
    const unsigned long long flat0 = ((unsigned long long)n * C + seg[q].c_first) * a.HW;
    if (VEC == 4) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      float4* o4 = reinterpret_cast<float4*>(o);
      for (int i = threadIdx.x; i < (len >> 2); i += 256) {
        const int c = seg[q].c_first + (i * 4) / a.HW;
        const float ga = a.gamma[c], be = a.beta[c];
        const float4 v = p4[i];
        float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u = ga * ((r[j] - mean) * rstd) + be;
          float t = a.act ? act_f(a.act, u) : u;
          if (a.drop_p > 0.f) t = stk_drop_field(stk_mix64(seed, (flat0 + (unsigned long long)(i * 4)) >> 2), j) >= a.drop_thr ? t * a.keep_scale : 0.f;
          r[j] = t;
        }
        o4[i] = make_float4(r[0], r[1], r[2], r[3]);
      }
    } else {
      for (int i = threadIdx.x; i < len; i += 256) {
        const int c = seg[q].c_first + i / a.HW;
        float u = a.gamma[c] * ((p[i] - mean) * rstd) + a.beta[c];
        float t = a.act ? act_f(a.act, u) : u;
        if (a.drop_p > 0.f) t = stk_keep(seed, flat0 + (unsigned long long)i, a.drop_thr) ? t * a.keep_scale : 0.f;
        o[i] = t;
      }
    }
  }
}

// ---- backward -------------------------------------------------------------------------------------
// du = dy * mask * act'(u);  dgamma_c = sum du*xhat;  dbeta_c = sum du;
// dx = rstd * (du*gamma - mean_g(du*gamma) - xhat * mean_g(du*gamma*xhat))
__device__ __forceinline__ float gn_du(const GnArgs& a, float xv, float dyv, float mean, float rstd, float ga, float be,
                                       unsigned long long seed, unsigned long long flat, float& xhat) {
  xhat = (xv - mean) * rstd;
  float go = dyv;
  if (a.drop_p > 0.f) go = stk_keep(seed, flat, a.drop_thr) ? go * a.keep_scale : 0.f;
  if (a.act) {
    const float u = ga * xhat + be;
    go = go * act_slope_f(a.act, u);
  }
  return go;
}

__global__ __launch_bounds__(256) void gn_bwd_kernel(GnArgs a, const float* __restrict__ dy,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     float* __restrict__ dx1, float beta1, float* __restrict__ dx2,
                                                     float beta2, float* __restrict__ ws) {
  __shared__ float red[16];
  const int ng = blockIdx.x;
  const int n = ng / a.G, g = ng - n * a.G;
  const int C = a.C1 + a.C2;
  const float mean = mean_in[ng], rstd = rstd_in[ng];
  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;

  // pass A: per-channel sums (channel loop; each channel is one contiguous HW run)
  float gs[2] = {0.f, 0.f};   // group sums of du*gamma and du*gamma*xhat (thread-partial)
  for (int c = g * a.cpg; c < (g + 1) * a.cpg; ++c) {
    const float* xp = c < a.C1 ? a.x1 + ((long)n * a.C1 + c) * a.HW : a.x2 + ((long)n * a.C2 + (c - a.C1)) * a.HW;
    const float* dp = dy + ((long)n * C + c) * a.HW;
    const unsigned long long flat0 = ((unsigned long long)n * C + c) * a.HW;
    const float ga = a.gamma[c], be = a.beta[c];
    float cs[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < a.HW; i += 256) {
      float xhat;
      const float du = gn_du(a, xp[i], dp[i], mean, rstd, ga, be, seed, flat0 + i, xhat);
      cs[0] += du;
      cs[1] += du * xhat;
    }
    gs[0] += cs[0] * ga;
    gs[1] += cs[1] * ga;
    block_sum<2>(cs, red);
    if (threadIdx.x == 0) {
      ws[((long)n * C + c) * 2 + 0] = cs[0];
      ws[((long)n * C + c) * 2 + 1] = cs[1];
    }
  }
  block_sum<2>(gs, red);
  const float inv_l = 1.f / ((float)a.cpg * (float)a.HW);
  const float m1 = gs[0] * inv_l, m2 = gs[1] * inv_l;

  // pass B: dx
  for (int c = g * a.cpg; c < (g + 1) * a.cpg; ++c) {
    const float* xp; float* op; float ob;
    if (c < a.C1) {
      xp = a.x1 + ((long)n * a.C1 + c) * a.HW;
      op = dx1 ? dx1 + ((long)n * a.C1 + c) * a.HW : nullptr;
      ob = beta1;
    } else {
      xp = a.x2 + ((long)n * a.C2 + (c - a.C1)) * a.HW;
      op = dx2 ? dx2 + ((long)n * a.C2 + (c - a.C1)) * a.HW : nullptr;
      ob = beta2;
    }
    if (!op) continue;
    const float* dp = dy + ((long)n * C + c) * a.HW;
    const unsigned long long flat0 = ((unsigned long long)n * C + c) * a.HW;
    const float ga = a.gamma[c], be = a.beta[c];
    for (int i = threadIdx.x; i < a.HW; i += 256) {
      float xhat;
      const float du = gn_du(a, xp[i], dp[i], mean, rstd, ga, be, seed, flat0 + i, xhat);
      const float r = rstd * (du * ga - m1 - xhat * m2);
      op[i] = (ob != 0.f ? ob * op[i] : 0.f) + r;
    }
  }
}

// Flat variant for the shapes the score network has: H*W a power of two >= 16, groups of at most 16384 elements,
// 16-byte aligned tensors.  The group is one flat run of float4 items, IPT per thread, and (du, xhat) stay in
// registers between the reduction and the dx write, so x and dy leave HBM once and the sigmoid is evaluated once
// (algorithmic traffic: read x, read dy, write dx).  Per-channel sums come from a segmented wave reduction (a wave
// covers one channel or 64/(HW/4) whole channels) written to fixed LDS slots and folded in slot order -- no
// atomics, so results are reproducible.  blockDim = 64/128/256 so small groups do not idle lanes.
template <int IPT, bool WANT>
__global__ __launch_bounds__(1024) void gn_bwd_flat_kernel(GnArgs a, const float* __restrict__ dy,
                                                          const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, float* __restrict__ dx1,
                                                          float beta1, float* __restrict__ dx2, float beta2,
                                                          float* __restrict__ ws, int hw_log2, GnBwdOut out) {
  __shared__ float s_part[2048], s_ch[1024], s_mx[16];
  const int T = blockDim.x;
  const int ng = blockIdx.x;
  const int n = ng / a.G, g = ng - n * a.G;
  const int C = a.C1 + a.C2;
  const float mean = mean_in[ng], rstd = rstd_in[ng];
  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;
  const int c0 = g * a.cpg;
  const int L4 = (a.cpg << hw_log2) >> 2;
  const int seglog = min(6, hw_log2 - 2);          // lanes of a wave that share a channel = 2^seglog
  const int lane = threadIdx.x & 63;

  float du[IPT][4], xh[IPT][4], gam[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int i = threadIdx.x + T * k;
    const bool valid = i < L4;
    const int e = valid ? 4 * i : 0;
    const int c = c0 + (e >> hw_log2), off = e & (a.HW - 1);
    const float* xp = c < a.C1 ? a.x1 + (((long)n * a.C1 + c) << hw_log2) + off
                               : a.x2 + (((long)n * a.C2 + (c - a.C1)) << hw_log2) + off;
    const unsigned long long flat = ((((unsigned long long)n * C + c)) << hw_log2) + off;
    const float4 xv = *reinterpret_cast<const float4*>(xp);
    const float4 dv = *reinterpret_cast<const float4*>(dy + flat);
    const float ga = a.gamma[c], be = a.beta[c];
    gam[k] = ga;
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
    float cs0 = 0.f, cs1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = gn_du(a, xs[j], ds[j], mean, rstd, ga, be, seed, (flat & ~3ULL) + j, xh[k][j]);
      du[k][j] = valid ? d : 0.f;
      cs0 += du[k][j];
      cs1 += du[k][j] * xh[k][j];
    }
    for (int o = 0; o < seglog; ++o) {
      cs0 += __shfl_xor(cs0, 1 << o);
      cs1 += __shfl_xor(cs1, 1 << o);
    }
    if (valid && (lane & ((1 << seglog) - 1)) == 0) {
      s_part[2 * (i >> seglog)] = cs0;
      s_part[2 * (i >> seglog) + 1] = cs1;
    }
  }
  __syncthreads();
  const int spc = (a.HW >> 2) >> seglog;            // slots per channel
  for (int cl = threadIdx.x; cl < a.cpg; cl += T) {
    float t0 = 0.f, t1 = 0.f;
    for (int q = 0; q < spc; ++q) {
      t0 += s_part[2 * (cl * spc + q)];
      t1 += s_part[2 * (cl * spc + q) + 1];
    }
    s_ch[2 * cl] = t0;
    s_ch[2 * cl + 1] = t1;
    ws[((long)n * C + c0 + cl) * 2 + 0] = t0;
    ws[((long)n * C + c0 + cl) * 2 + 1] = t1;
  }
  __syncthreads();
  float g0 = 0.f, g1 = 0.f;
  for (int cl = 0; cl < a.cpg; ++cl) {
    const float ga = a.gamma[c0 + cl];
    g0 += ga * s_ch[2 * cl];
    g1 += ga * s_ch[2 * cl + 1];
  }
  const float inv_l = 1.f / ((float)a.cpg * (float)a.HW);
  const float m1 = g0 * inv_l, m2 = g1 * inv_l;

  // `out` (stk_gn_bwd_out_f32, single-source layers): what the consumer of dx1 -- the backward of the convolution that
  // produced x1 -- would otherwise take one more pass over dx1 for: per-(sample, channel) sums of the FINAL dx1 values
  // (bias / time-embedding gradients) and max |dx1| (scale record of its planes).  Block-uniform branches only.
  constexpr bool want = WANT;            // compile-time: the plain instances carry none of the by-product code
  float amax_l = 0.f;
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int i = threadIdx.x + T * k;
    const bool valid = i < L4;
    const int e = valid ? 4 * i : 0;
    const int c = c0 + (e >> hw_log2), off = e & (a.HW - 1);
    float* op; float ob;
    if (c < a.C1) { op = dx1 ? dx1 + (((long)n * a.C1 + c) << hw_log2) + off : nullptr; ob = beta1; }
    else { op = dx2 ? dx2 + (((long)n * a.C2 + (c - a.C1)) << hw_log2) + off : nullptr; ob = beta2; }
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid && op) {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = rstd * (du[k][j] * gam[k] - m1 - xh[k][j] * m2);
      if (ob != 0.f) {
        const float4 old = *reinterpret_cast<const float4*>(op);
        r[0] += ob * old.x; r[1] += ob * old.y; r[2] += ob * old.z; r[3] += ob * old.w;
      }
      if (want && out.add && c < a.C1) {
        const float4 ad = *reinterpret_cast<const float4*>(out.add + (((long)n * a.C1 + c) << hw_log2) + off);
        r[0] += out.add_scale * ad.x; r[1] += out.add_scale * ad.y; r[2] += out.add_scale * ad.z; r[3] += out.add_scale * ad.w;
      }
      *reinterpret_cast<float4*>(op) = make_float4(r[0], r[1], r[2], r[3]);
    }
    if (want) {
      const bool mine = c < a.C1;                           // by-products cover dx1 only
      float rs = mine ? (r[0] + r[1]) + (r[2] + r[3]) : 0.f;
      if (mine) amax_l = fmaxf(amax_l, fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fmaxf(fabsf(r[2]), fabsf(r[3]))));
      for (int o = 0; o < seglog; ++o) rs += __shfl_xor(rs, 1 << o);
      if (valid && (lane & ((1 << seglog) - 1)) == 0) s_part[i >> seglog] = rs;     // s_part is free after the barrier above
    }
  }
  if (want) {
    {
      const float m = wave_max(amax_l);
      if (lane == 0) s_mx[threadIdx.x >> 6] = m;          // (not s_ch: another wave may still be reading its group sums)
    }
    __syncthreads();
    for (int cl = threadIdx.x; cl < a.cpg; cl += T) {
      if (c0 + cl >= a.C1) continue;
      float t = 0.f;
      for (int q = 0; q < spc; ++q) t += s_part[cl * spc + q];
      t *= out.scale;
      if (out.sum) { out.sum[((long)n * a.C1 + c0 + cl) * 2] = t; out.sum[((long)n * a.C1 + c0 + cl) * 2 + 1] = t; }
      if (out.temb) out.temb[(long)n * out.temb_stride + c0 + cl] = t;
    }
    if (out.amax && threadIdx.x == 0) {
      // ONE atomic per workgroup (4096 workgroups on 256 slots: 16 per address; one per wave measured no gain over the
      // separate pass).  Non-negative floats order like their bit patterns: an integer max is exact and order-independent
      float m = 0.f;
      for (int w = 0; w < (T >> 6); ++w) m = fmaxf(m, s_mx[w]);
      // (slot = block id mod 256, NOT image mod 256: the 32 workgroups of an image run at the same time, and 32 atomics on one
      // address cost this kernel 44 -> 54 us on the 32x32 layers, 11 -> 18 us on the 8x8 ones -- measured, round 5)
      atomicMax(reinterpret_cast<unsigned*>(out.amax) + (blockIdx.x & 255), __float_as_uint(m));
    }
  }
}

// ---- large groups: one (sample, group) split over many workgroups -------------------------------------------
// A 256x256 map at batch 4 has only N*G = 128 groups of 1 MB each: one workgroup per group leaves half the chip
// idle and streams each group serially (measured 689 us for a 134 MB backward).  Here a workgroup owns one CHUNK
// (GN_CHUNK consecutive elements of one channel); a first kernel writes per-chunk partial sums, the second one
// folds the partials of its group in a fixed order and streams its chunk.  Partials: part[((n*C + c)*Sc + k)*2 + {0,1}].
constexpr int GN_CHUNK = 4096;       // floats: 4 float4 per thread

__device__ __forceinline__ const float* gn_chan(const GnArgs& a, int n, int c) {
  return c < a.C1 ? a.x1 + ((long)n * a.C1 + c) * a.HW : a.x2 + ((long)n * a.C2 + (c - a.C1)) * a.HW;
}

// forward statistics: shifted sums (shift = first element of the group) of one chunk
__global__ __launch_bounds__(256) void gn_split_stats_kernel(GnArgs a, float* __restrict__ part, int Sc) {
  __shared__ float red[16];
  const int k = blockIdx.x % Sc, nc = blockIdx.x / Sc;
  const int C = a.C1 + a.C2, n = nc / C, c = nc - n * C, g = c / a.cpg;
  const float shift = gn_chan(a, n, g * a.cpg)[0];
  const float4* p4 = reinterpret_cast<const float4*>(gn_chan(a, n, c) + (long)k * GN_CHUNK);
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GN_CHUNK / 1024; ++i) {
    const float4 v = p4[threadIdx.x + 256 * i];
    const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
    s[0] += (d0 + d1) + (d2 + d3);
    s[1] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  block_sum<2>(s, red);
  if (threadIdx.x == 0) { part[(long)blockIdx.x * 2] = s[0]; part[(long)blockIdx.x * 2 + 1] = s[1]; }
}

__global__ __launch_bounds__(256) void gn_split_fwd_kernel(GnArgs a, const float* __restrict__ part, int Sc,
                                                           float* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, float eps) {
  __shared__ float red[16];
  // chunks in the REVERSE of the order the statistics kernel read them in: what that pass read last is still in the
  // Infinity Cache (256 MB) -- in the same order a tensor of about that size evicts each piece just before its reuse
  const int bid = gridDim.x - 1 - blockIdx.x;
  const int k = bid % Sc, nc = bid / Sc;
  const int C = a.C1 + a.C2, n = nc / C, c = nc - n * C, g = c / a.cpg;
  const float shift = gn_chan(a, n, g * a.cpg)[0];
  // fold the cpg * Sc partials of the group (fixed order inside block_sum)
  float s[2] = {0.f, 0.f};
  const long pbase = ((long)n * C + g * a.cpg) * Sc;
  for (int i = threadIdx.x; i < a.cpg * Sc; i += 256) { s[0] += part[(pbase + i) * 2]; s[1] += part[(pbase + i) * 2 + 1]; }
  block_sum<2>(s, red);
  const float inv_l = 1.f / ((float)a.cpg * (float)a.HW);
  const float md = s[0] * inv_l;
  const float var = fmaxf(s[1] * inv_l - md * md, 0.f);
  const float mean = shift + md, rstd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0 && k == 0 && c == g * a.cpg) { mean_out[n * a.G + g] = mean; rstd_out[n * a.G + g] = rstd; }
  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;
  const float ga = a.gamma[c], be = a.beta[c];
  const long coff = (long)k * GN_CHUNK;
  const float4* p4 = reinterpret_cast<const float4*>(gn_chan(a, n, c) + coff);
  float4* o4 = reinterpret_cast<float4*>(y + ((long)n * C + c) * a.HW + coff);
  const unsigned long long flat0 = ((unsigned long long)n * C + c) * a.HW + coff;
#pragma unroll
  for (int i = 0; i < GN_CHUNK / 1024; ++i) {
    const int e = threadIdx.x + 256 * i;
    const float4 v = p4[e];
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = ga * ((r[j] - mean) * rstd) + be;
      float t = a.act ? act_f(a.act, u) : u;
      if (a.drop_p > 0.f) t = stk_drop_field(stk_mix64(seed, (flat0 + (unsigned long long)(e * 4)) >> 2), j) >= a.drop_thr ? t * a.keep_scale : 0.f;
      r[j] = t;
    }
    o4[e] = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// backward partials of one chunk: sum du, sum du * xhat
__global__ __launch_bounds__(256) void gn_split_bwd_part_kernel(GnArgs a, const float* __restrict__ dy,
                                                                const float* __restrict__ mean_in,
                                                                const float* __restrict__ rstd_in,
                                                                float* __restrict__ part, int Sc) {
  __shared__ float red[16];
  const int k = blockIdx.x % Sc, nc = blockIdx.x / Sc;
  const int C = a.C1 + a.C2, n = nc / C, c = nc - n * C, g = c / a.cpg;
  const float mean = mean_in[n * a.G + g], rstd = rstd_in[n * a.G + g];
  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;
  const float ga = a.gamma[c], be = a.beta[c];
  const long coff = (long)k * GN_CHUNK;
  const float4* x4 = reinterpret_cast<const float4*>(gn_chan(a, n, c) + coff);
  const float4* d4 = reinterpret_cast<const float4*>(dy + ((long)n * C + c) * a.HW + coff);
  const unsigned long long flat0 = ((unsigned long long)n * C + c) * a.HW + coff;
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GN_CHUNK / 1024; ++i) {
    const int e = threadIdx.x + 256 * i;
    const float4 xv = x4[e], dv = d4[e];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float xh;
      const float du = gn_du(a, xs[j], ds[j], mean, rstd, ga, be, seed, ((flat0 + (unsigned long long)(e * 4)) & ~3ULL) + j, xh);
      s[0] += du;
      s[1] += du * xh;
    }
  }
  block_sum<2>(s, red);
  if (threadIdx.x == 0) { part[(long)blockIdx.x * 2] = s[0]; part[(long)blockIdx.x * 2 + 1] = s[1]; }
}

__global__ __launch_bounds__(256) void gn_split_bwd_kernel(GnArgs a, const float* __restrict__ dy,
                                                           const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, const float* __restrict__ part,
                                                           int Sc, float* __restrict__ dx1, float beta1,
                                                           float* __restrict__ dx2, float beta2, float* __restrict__ ws) {
  __shared__ float red[16];
  const int bid = gridDim.x - 1 - blockIdx.x;          // reverse of gn_split_bwd_part_kernel's order (see gn_split_fwd_kernel)
  const int k = bid % Sc, nc = bid / Sc;
  const int C = a.C1 + a.C2, n = nc / C, c = nc - n * C, g = c / a.cpg;
  const float mean = mean_in[n * a.G + g], rstd = rstd_in[n * a.G + g];
  // group sums of du*gamma and du*gamma*xhat from the partials (fixed order), and this channel's own sums
  float s[2] = {0.f, 0.f};
  const long pbase = ((long)n * C + g * a.cpg) * Sc;
  for (int i = threadIdx.x; i < a.cpg * Sc; i += 256) {
    const float gc = a.gamma[g * a.cpg + i / Sc];
    s[0] += gc * part[(pbase + i) * 2];
    s[1] += gc * part[(pbase + i) * 2 + 1];
  }
  block_sum<2>(s, red);
  const float inv_l = 1.f / ((float)a.cpg * (float)a.HW);
  const float m1 = s[0] * inv_l, m2 = s[1] * inv_l;
  if (k == 0 && threadIdx.x == 0) {            // per-(sample, channel) sums for the parameter gradients
    float t0 = 0.f, t1 = 0.f;
    const long cb = ((long)n * C + c) * Sc;
    for (int q = 0; q < Sc; ++q) { t0 += part[(cb + q) * 2]; t1 += part[(cb + q) * 2 + 1]; }
    ws[((long)n * C + c) * 2] = t0;
    ws[((long)n * C + c) * 2 + 1] = t1;
  }
  float* op; float ob;
  if (c < a.C1) { op = dx1 ? dx1 + ((long)n * a.C1 + c) * a.HW : nullptr; ob = beta1; }
  else { op = dx2 ? dx2 + ((long)n * a.C2 + (c - a.C1)) * a.HW : nullptr; ob = beta2; }
  if (!op) return;
  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;
  const float ga = a.gamma[c], be = a.beta[c];
  const long coff = (long)k * GN_CHUNK;
  const float4* x4 = reinterpret_cast<const float4*>(gn_chan(a, n, c) + coff);
  const float4* d4 = reinterpret_cast<const float4*>(dy + ((long)n * C + c) * a.HW + coff);
  float4* o4 = reinterpret_cast<float4*>(op + coff);
  const unsigned long long flat0 = ((unsigned long long)n * C + c) * a.HW + coff;
#pragma unroll
  for (int i = 0; i < GN_CHUNK / 1024; ++i) {
    const int e = threadIdx.x + 256 * i;
    const float4 xv = x4[e], dv = d4[e];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float xh;
      const float du = gn_du(a, xs[j], ds[j], mean, rstd, ga, be, seed, ((flat0 + (unsigned long long)(e * 4)) & ~3ULL) + j, xh);
      r[j] = rstd * (du * ga - m1 - xh * m2);
    }
    if (ob != 0.f) {
      const float4 old = o4[e];
      r[0] += ob * old.x; r[1] += ob * old.y; r[2] += ob * old.z; r[3] += ob * old.w;
    }
    o4[e] = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// Split path preconditions: groups too large for the register-resident kernels, channels made of whole chunks,
// 16-byte aligned tensors.
inline bool gn_split_ok(int HW, int cpg) { return (long)cpg * HW > 16384 && HW % GN_CHUNK == 0; }

// dgamma[c] += sum_n ws[n,c,1];  dbeta[c] += sum_n ws[n,c,0].  32 channels per block, 8-way split over n.
// With a descriptor table, blockIdx.y picks the layer (stk_gn_param_grad_batch).
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int N, int C,
                                                            const StkGnFoldDesc* __restrict__ descs) {
  __shared__ float redb[256], redg[256];
  if (descs) {
    const StkGnFoldDesc d = descs[blockIdx.y];
    ws = d.part; dgamma = d.dgamma; dbeta = d.dbeta; N = d.N; C = d.C;
    if ((int)blockIdx.x * 32 >= C) return;             // block-uniform: a layer narrower than the widest one
  }
  const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float sb = 0.f, sg = 0.f;
  if (c < C)
    for (int n = part; n < N; n += 8) {
      const float2 v = *reinterpret_cast<const float2*>(ws + ((long)n * C + c) * 2);
      sb += v.x;
      sg += v.y;
    }
  redb[threadIdx.x] = sb;
  redg[threadIdx.x] = sg;
  __syncthreads();
  if (part == 0 && c < C) {
    float tb = 0.f, tg = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { tb += redb[q * 32 + cl]; tg += redg[q * 32 + cl]; }
    if (dgamma) dgamma[c] += tg;
    if (dbeta) dbeta[c] += tb;
  }
}

}  // namespace

// A-priori bound of |GroupNorm(+SiLU)(+dropout) output| from the affine parameters alone, written as a planes scale
// record (include/stk.h "Planes"): rec[0] = bound, rec[1..255] = 0.  A group of L elements has |xhat| <= sqrt(L - 1)
// whatever the data, |silu(u)| <= |u|, dropout multiplies by 1 / (1 - p):
//     |y| <= (max|gamma| sqrt(L - 1) + max|beta|) / (1 - p).
// The consumer convolution scales its split by this bound instead of a measured maximum, which removes the |x| pass over
// every GroupNorm output; the bound is loose (sqrt(L - 1) = 64 for a 4 x 32 x 32 group against a typical |xhat| <= 5),
// which costs the split's second term a few of its 11 spare bits and nothing else (tests/test_planes.py).
__global__ __launch_bounds__(256) void gn_bound_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                       float sqrt_lm1, float keep_scale, float* __restrict__ rec) {
  __shared__ float red[8];
  float g = 0.f, b = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    g = fmaxf(g, fabsf(gamma[c]));
    b = fmaxf(b, fabsf(beta[c]));
  }
  g = wave_max(g); b = wave_max(b);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = g; red[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  g = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  b = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  rec[threadIdx.x] = threadIdx.x == 0 ? __fmaf_rn(g, sqrt_lm1, b) * keep_scale : 0.f;
}

// ---- forward writing planes --------------------------------------------------------------------------------------
// GroupNorm (+SiLU) (+dropout) whose output goes to the split convolutions as planes (include/stk.h "Planes"):
// [split][n][c / 32][pixel][c % 32] fp16.  One workgroup per (sample, 32-channel block) -- a whole number of groups when
// 32 % (C / G) == 0 -- and a thread per (pixel, 8-channel piece): its 8 loads are channel-strided but 16 lanes x 4 B =
// 64 B contiguous along pixels, and its two 16-byte stores (hi, lo) land lane-contiguously, 1 KB per wave and plane.
// The block (32 x HW <= 32768 elements) stays in registers between the statistics and the normalisation, so x is read
// once; the fp32 NCHW copy of y is optional (the weight gradient and non-split consumers read it; a no-grad pass whose
// consumers all take planes skips it).  The planes' scale comes from the a-priori bound of gn_bound_kernel, computed
// here from gamma / beta by every workgroup (C <= 1024 values) -- no pass over y, no dependence between workgroups.
template <int PASSES>
__global__ __launch_bounds__(1024) void gn_fwd_pl_kernel(GnArgs a, float* __restrict__ y, unsigned char* __restrict__ planes,
                                                         long plane_stride, float* __restrict__ rec,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         float eps, float sqrt_lm1, float* __restrict__ xmax1,
                                                         float* __restrict__ xmax2) {
  __shared__ float red[16 * 4 * 4];       // [wave][q][lo sum, lo sumsq, hi sum, hi sumsq]
  __shared__ float xm[16];                // per-wave max |x| (xmax by-product)
  __shared__ float gst[8 * 2];            // [group in block][mean, rstd]
  __shared__ float bnd[32];
  const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
  const int C = a.C1 + a.C2, Cb = C >> 5;
  const int n = blockIdx.x / Cb, cb = blockIdx.x - n * Cb;
  const int c0 = cb * 32;
  const float* src = c0 < a.C1 ? a.x1 + ((long)n * a.C1 + c0) * a.HW : a.x2 + ((long)n * a.C2 + (c0 - a.C1)) * a.HW;
  const int q = tid & 3;                  // 8-channel piece (constant per thread: T % 4 == 0)
  const int gb = 32 / a.cpg;              // groups in this block
  const int g_lo = (8 * q) / a.cpg, g_hi = (8 * q + 4) / a.cpg;
  // shifted sums (shift = first element of the group), as in gn_fwd_flat_kernel
  const float sh_lo = src[(long)(g_lo * a.cpg) * a.HW], sh_hi = src[(long)(g_hi * a.cpg) * a.HW];

  // scale of the planes: |y| <= (max|gamma| sqrt(L - 1) + max|beta|) / (1 - p)
  float gm = 0.f, bm = 0.f;
  for (int c = tid; c < C; c += T) { gm = fmaxf(gm, fabsf(a.gamma[c])); bm = fmaxf(bm, fabsf(a.beta[c])); }
  gm = wave_max(gm); bm = wave_max(bm);
  if (lane == 0) { bnd[wave] = gm; bnd[16 + wave] = bm; }

  float v[PASSES][8];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < PASSES; ++k) {
    const int px = (tid + T * k) >> 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[k][j] = src[(long)(8 * q + j) * a.HW + px];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d0 = v[k][j] - sh_lo, d1 = v[k][4 + j] - sh_hi;
      s[0] += d0; s[1] += d0 * d0; s[2] += d1; s[3] += d1 * d1;
    }
  }
  // lanes with equal q (lane bits 0..1) -> xor-shuffle over lane bits 2..5
#pragma unroll
  for (int o = 4; o < 64; o <<= 1)
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] += __shfl_xor(s[i], o, 64);
  if (lane < 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * 4 + lane) * 4 + i] = s[i];
  }
  if (xmax1) {
    // by-product for the block's 1x1 shortcut convolution, which reads the same source tensors as fp32 operands of the
    // split kernels: max |x| of this block, by atomic maximum into the 256-slot scale record of its source (zeroed by the
    // caller; non-negative floats order like their bit patterns)
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < PASSES; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[k][j]));
    m = wave_max(m);
    if (lane == 0) xm[wave] = m;
  }
  __syncthreads();
  if (xmax1 && tid == 0) {
    float m = 0.f;
    for (int w = 0; w < nw; ++w) m = fmaxf(m, xm[w]);
    float* rec = c0 < a.C1 ? xmax1 : xmax2;
    if (rec) atomicMax(reinterpret_cast<unsigned*>(rec) + (blockIdx.x & 255), __float_as_uint(m));
  }
  if (tid < gb) {
    // pieces of group `tid`, in a fixed order: (q', half) with (8 q' + 4 half) / cpg == tid, over all waves
    float s0 = 0.f, s1 = 0.f;
    for (int qq = 0; qq < 4; ++qq)
      for (int h = 0; h < 2; ++h)
        if ((8 * qq + 4 * h) / a.cpg == tid)
          for (int w = 0; w < nw; ++w) { s0 += red[(w * 4 + qq) * 4 + 2 * h]; s1 += red[(w * 4 + qq) * 4 + 2 * h + 1]; }
    const float inv_l = 1.f / ((float)a.cpg * (float)a.HW);
    const float shift = src[(long)(tid * a.cpg) * a.HW];
    const float md = s0 * inv_l;
    const float var = fmaxf(s1 * inv_l - md * md, 0.f);
    const float mean = shift + md, rstd = 1.f / sqrtf(var + eps);
    gst[2 * tid] = mean; gst[2 * tid + 1] = rstd;
    const int g = (c0 / a.cpg) + tid;
    mean_out[n * a.G + g] = mean;
    rstd_out[n * a.G + g] = rstd;
  }
  __syncthreads();
  float gmax = 0.f, bmax = 0.f;
  for (int w = 0; w < nw; ++w) { gmax = fmaxf(gmax, bnd[w]); bmax = fmaxf(bmax, bnd[16 + w]); }
  const float bound = __fmaf_rn(gmax, sqrt_lm1, bmax) * a.keep_scale;
  if (blockIdx.x == 0)
    for (int i = tid; i < 256; i += T) rec[i] = i == 0 ? bound : 0.f;
  // the power of two that puts the bound in [2^13, 2^14) (x2::pow2_scale_of in conv_x2.h)
  float sc = 1.f;
  {
    const int be = (int)((__float_as_uint(bound) >> 23) & 0xffu);
    if (be != 0) sc = __uint_as_float((unsigned)min(max(127 + 13 - (be - 127), 1), 254) << 23);
  }
  const float m_lo = gst[2 * g_lo], r_lo = gst[2 * g_lo + 1], m_hi = gst[2 * g_hi], r_hi = gst[2 * g_hi + 1];
  float ga[8], be8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ga[j] = a.gamma[c0 + 8 * q + j]; be8[j] = a.beta[c0 + 8 * q + j]; }
  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;
  float* yo = y ? y + ((long)n * C + c0) * a.HW : nullptr;
  unsigned char* po = planes + ((long)n * Cb + cb) * a.HW * 64 + q * 16;
#pragma unroll
  for (int k = 0; k < PASSES; ++k) {
    const int px = (tid + T * k) >> 2;
    // Dropout: the four lanes of a row that hold pixels 4 m .. 4 m + 3 of the same 8-channel piece (lanes 4 apart) need
    // the same eight RNG quads (one per channel, stk_rng.h).  Lane r of the four computes the two mixes of channels 2 r,
    // 2 r + 1 and the others fetch their 16-bit fields by lane shuffle: 2 mixes per thread and pass instead of 8.
    unsigned keep = 0xffu;
    if (a.drop_p > 0.f) {
      const int r = (tid >> 2) & 3;                              // = px & 3: T / 4 is a multiple of 4
      const unsigned long long fa = ((unsigned long long)n * C + (c0 + 8 * q + 2 * r)) * a.HW + px;
      const unsigned long long za = stk_mix64(seed, fa >> 2), zb = stk_mix64(seed, (fa + a.HW) >> 2);
      keep = 0u;
#pragma unroll
      for (int rot = 0; rot < 4; ++rot) {
        // round `rot`: lane r reads from the lane that owns channel pair pr = (r + rot) & 3; seen from the sender, its
        // reader is lane (r - rot) & 3, so it sends that lane's two 16-bit fields packed into one word
        const unsigned rd = (unsigned)((r - rot) & 3);
        const unsigned send = stk_drop_field(za, rd) | (stk_drop_field(zb, rd) << 16);
        const int pr = (r + rot) & 3;
        const unsigned got = rot == 0 ? send : (unsigned)__shfl((int)send, (lane & ~12) | (pr << 2), 64);
        keep |= ((got & 0xffffu) >= a.drop_thr ? 1u : 0u) << (2 * pr);
        keep |= ((got >> 16) >= a.drop_thr ? 1u : 0u) << (2 * pr + 1);
      }
    }
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float mean = j < 4 ? m_lo : m_hi, rstd = j < 4 ? r_lo : r_hi;
      const float u = ga[j] * ((v[k][j] - mean) * rstd) + be8[j];
      float r = a.act ? act_f(a.act, u) : u;
      if (a.drop_p > 0.f) r = ((keep >> j) & 1u) ? r * a.keep_scale : 0.f;
      t[j] = r;
      if (yo) yo[(long)(8 * q + j) * a.HW + px] = r;
    }
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = sc * t[2 * j], v1 = sc * t[2 * j + 1];
      const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
      const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      hi[j] = __builtin_bit_cast(unsigned, h2{h0, h1});
      lo[j] = __builtin_bit_cast(unsigned, h2{l0, l1});
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<u4*>(po + (long)px * 64) = u4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u4*>(po + plane_stride + (long)px * 64) = u4{lo[0], lo[1], lo[2], lo[3]};
  }
}

// ---- forward writing planes, two kernels (round 3) -----------------------------------------------------------------
// The one-pass kernel above holds a (sample, 32-channel block) in registers between the statistics and the
// normalisation: on the large maps that is one or two big workgroups per CU whose load, reduce and store phases do not
// overlap (32x32: 2.7-3.3 TB/s against 6.9 for a copy).  Here the statistics are their own streaming kernel (x read once,
// nothing written but mean / rstd), and the normalisation + SiLU + dropout + split is a second one with the tiling of
// pl::split_planes_kernel -- a workgroup per (sample, 32-channel block, 128-pixel tile), all element-wise work done on
// the float4 (four consecutive pixels of one channel = one dropout RNG quad) before the tile goes through LDS for the
// transposition to [pixel][32 channels].  x is read twice, the second time out of the Infinity Cache.
__global__ __launch_bounds__(256) void gn_stats_kernel(GnArgs a, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       float eps, float sqrt_lm1, float* __restrict__ rec) {
  __shared__ float red[16];
  const int ng = blockIdx.x;
  const int n = ng / a.G, g = ng - n * a.G;
  Seg seg[2];
  group_segments(a, n, g, seg);                          // whole groups per source (C1 % cpg == 0): one segment is empty
  const float* p = seg[0].len ? seg[0].p : seg[1].p;
  const int L = a.cpg * a.HW, L4 = L >> 2;
  const float shift = p[0];
  const float4* p4 = reinterpret_cast<const float4*>(p);
  float s[2] = {0.f, 0.f};
  for (int i = threadIdx.x; i < L4; i += 256) {
    const float4 v = p4[i];
    const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
    s[0] += (d0 + d1) + (d2 + d3);
    s[1] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  block_sum<2>(s, red);
  const float inv_l = 1.f / (float)L;
  const float md = s[0] * inv_l;
  const float var = fmaxf(s[1] * inv_l - md * md, 0.f);
  const float rstd_v = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[ng] = shift + md;
    rstd_out[ng] = rstd_v;
  }
  if (rec && ng == 0) {                                  // the planes' scale record: a-priori bound of |y| (gn_bound_kernel)
    const int C = a.C1 + a.C2;
    float gm = 0.f, bm = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { gm = fmaxf(gm, fabsf(a.gamma[c])); bm = fmaxf(bm, fabsf(a.beta[c])); }
    gm = wave_max(gm); bm = wave_max(bm);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = gm; red[4 + (threadIdx.x >> 6)] = bm; }
    __syncthreads();
    gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    bm = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    rec[threadIdx.x] = threadIdx.x == 0 ? __fmaf_rn(gm, sqrt_lm1, bm) * a.keep_scale : 0.f;
  }
}

// mean / rstd of large groups from the per-chunk partials of gn_split_stats_kernel (fixed order), + the scale record
__global__ __launch_bounds__(256) void gn_fold_stats_kernel(GnArgs a, const float* __restrict__ part, int Sc,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            float eps) {
  __shared__ float red[16];
  const int ng = blockIdx.x;
  const int n = ng / a.G, g = ng - n * a.G;
  const int C = a.C1 + a.C2;
  const float shift = gn_chan(a, n, g * a.cpg)[0];
  float s[2] = {0.f, 0.f};
  const long pbase = ((long)n * C + g * a.cpg) * Sc;
  for (int i = threadIdx.x; i < a.cpg * Sc; i += 256) { s[0] += part[(pbase + i) * 2]; s[1] += part[(pbase + i) * 2 + 1]; }
  block_sum<2>(s, red);
  const float inv_l = 1.f / ((float)a.cpg * (float)a.HW);
  const float md = s[0] * inv_l;
  const float var = fmaxf(s[1] * inv_l - md * md, 0.f);
  if (threadIdx.x == 0) { mean_out[ng] = shift + md; rstd_out[ng] = 1.f / sqrtf(var + eps); }
}

constexpr int AP_PIX = 128;
__global__ __launch_bounds__(256) void gn_apply_pl_kernel(GnArgs a, const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, const float* __restrict__ rec,
                                                          float* __restrict__ y, unsigned char* __restrict__ planes,
                                                          long plane_stride, float* __restrict__ xmax1,
                                                          float* __restrict__ xmax2) {
  __shared__ float t[32 * 129];
  __shared__ float xm[4];
  const int tid = threadIdx.x;
  const int C = a.C1 + a.C2, Cb = C >> 5;
  const int tiles = a.HW / AP_PIX;
  const int tile = blockIdx.x % tiles, rest = blockIdx.x / tiles;
  const int cb = rest % Cb, n = rest / Cb;
  const int c0 = cb * 32, p0 = tile * AP_PIX;
  const float* src = c0 < a.C1 ? a.x1 + ((long)n * a.C1 + c0) * a.HW : a.x2 + ((long)n * a.C2 + (c0 - a.C1)) * a.HW;
  // the power of two that puts the bound in [2^13, 2^14) (x2::pow2_scale_of in conv_x2.h)
  float sc = 1.f;
  {
    const int be = (int)((__float_as_uint(rec[0]) >> 23) & 0xffu);
    if (be != 0) sc = __uint_as_float((unsigned)min(max(127 + 13 - (be - 127), 1), 254) << 23);
  }
  unsigned long long seed = a.seed;
  if (a.drop_p > 0.f && a.seed_dev) seed += *a.seed_dev;
  float amax_l = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + 256 * k;
    const int c = i >> 5, q4 = i & 31;
    const int ch = c0 + c, px = p0 + 4 * q4;
    const float4 v = *reinterpret_cast<const float4*>(src + (long)c * a.HW + px);
    const int g = ch / a.cpg;
    const float mean = mean_in[n * a.G + g], rstd = rstd_in[n * a.G + g];
    const float ga = a.gamma[ch], be = a.beta[ch];
    float r[4] = {v.x, v.y, v.z, v.w};
    if (xmax1) amax_l = fmaxf(amax_l, fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fmaxf(fabsf(r[2]), fabsf(r[3]))));
    const unsigned long long flat = ((unsigned long long)n * C + ch) * a.HW + px;
    unsigned long long z = 0;
    if (a.drop_p > 0.f) z = stk_mix64(seed, flat >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = ga * ((r[j] - mean) * rstd) + be;
      float o = a.act ? act_f(a.act, u) : u;
      if (a.drop_p > 0.f) o = stk_drop_field(z, j) >= a.drop_thr ? o * a.keep_scale : 0.f;
      r[j] = o;
    }
    if (y) *reinterpret_cast<float4*>(y + ((long)n * C + ch) * a.HW + px) = make_float4(r[0], r[1], r[2], r[3]);
    float* d = t + c * 129 + 4 * q4;
    d[0] = sc * r[0]; d[1] = sc * r[1]; d[2] = sc * r[2]; d[3] = sc * r[3];
  }
  if (xmax1) {
    const float m = wave_max(amax_l);
    if ((tid & 63) == 0) xm[tid >> 6] = m;
  }
  __syncthreads();
  if (xmax1 && tid == 0) {
    float* dst = c0 < a.C1 ? xmax1 : xmax2;
    if (dst) atomicMax(reinterpret_cast<unsigned*>(dst) + (blockIdx.x & 255),
                       __float_as_uint(fmaxf(fmaxf(xm[0], xm[1]), fmaxf(xm[2], xm[3]))));
  }
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int id = tid + 256 * r2;
    const int px = id >> 2, q = id & 3;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = t[(8 * q + 2 * j) * 129 + px], v1 = t[(8 * q + 2 * j + 1) * 129 + px];
      const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
      const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      hi[j] = __builtin_bit_cast(unsigned, h2{h0, h1});
      lo[j] = __builtin_bit_cast(unsigned, h2{l0, l1});
    }
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    unsigned char* o = planes + (((long)n * Cb + cb) * a.HW + p0 + px) * 64 + q * 16;
    *reinterpret_cast<u4*>(o) = u4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<u4*>(o + plane_stride) = u4{lo[0], lo[1], lo[2], lo[3]};
  }
}

// shapes of the two-kernel route: maps of whole 128-pixel tiles, whole groups per 32-channel block and per source
static inline bool gn_pl_2k_ok(int C1, int C2, int HW, int G) {
  const int C = C1 + C2;
  if (C % 32 || C1 % 32 || C % G || HW % AP_PIX) return false;
  const int cpg = C / G;
  if (cpg > 32 || 32 % cpg) return false;
  const long L = (long)cpg * HW;
  return L <= 16384 || gn_split_ok(HW, cpg);
}

// shapes the fused kernel takes: whole groups per 32-channel block, both sources in whole blocks, the block in registers
static inline bool gn_pl_fused_ok(int C1, int C2, int HW, int G) {
  const int C = C1 + C2;
  if (C % 32 || C1 % 32 || C % G) return false;
  const int cpg = C / G;
  if (cpg > 32 || 32 % cpg || cpg < 4) return false;        // a thread's 4-channel halves must not straddle groups
  return HW == 16 || HW == 64 || HW == 256 || HW == 1024;
}

extern "C" {

int stk_gn_bound_f32(const float* gamma, const float* beta, int C, int G, int HW, float drop_p, float* rec, void* stream) {
  if (!gamma || !beta || !rec || C <= 0 || G <= 0 || C % G || HW <= 0 || drop_p < 0.f || drop_p >= 1.f) return STK_EINVAL;
  const float L = (float)((long)(C / G) * HW);
  hipLaunchKernelGGL(gn_bound_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gamma, beta, C, sqrtf(L - 1.f),
                     1.f / (1.f - drop_p), rec);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_gn_fwd_f32(const float* x1, int C1, const float* x2, int C2, const float* gamma, const float* beta, float* y,
                   float* mean, float* rstd, int N, int HW, int G, float eps, int act, float drop_p,
                   unsigned long long seed, const unsigned long long* seed_dev, float* ws, void* stream) {
  const int C = C1 + C2;
  if (!x1 || !gamma || !beta || !y || !mean || !rstd || N <= 0 || HW <= 0 || G <= 0 || C1 <= 0 || C2 < 0 || C % G ||
      (C2 > 0 && !x2) || drop_p < 0.f || drop_p >= 1.f || act < 0 || act > STK_ACT_ELU)
    return STK_EINVAL;
  GnArgs a;
  a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.gamma = gamma; a.beta = beta;
  a.N = N; a.HW = HW; a.G = G; a.cpg = C / G; a.act = act; a.drop_p = drop_p; a.keep_scale = 1.f / (1.f - drop_p); a.drop_thr = stk_drop_threshold(drop_p);
  a.seed = seed; a.seed_dev = seed_dev;
  const bool vec = (HW & 3) == 0 && stk_aligned16(x1) && stk_aligned16(y) && (!x2 || stk_aligned16(x2));
  if (ws && vec && gn_split_ok(HW, a.cpg)) {
    const int Sc = HW / GN_CHUNK;
    const dim3 grid((unsigned)((long)N * C * Sc));
    hipLaunchKernelGGL(gn_split_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, ws, Sc);
    STK_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_split_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, ws, Sc, y, mean, rstd, eps);
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
  const long L = (long)a.cpg * HW;
  if (vec && L <= 16384) {
    const int L4 = (int)(L >> 2);
    int T = 64;
    while (T < 1024 && T * 4 < L4) T <<= 1;
    const int ipt = stk_cdiv(L4, T);
#define STK_GN_FWD_FLAT(IPT)                                                                                     \
  hipLaunchKernelGGL((gn_fwd_flat_kernel<IPT>), dim3(N * G), dim3(T), 0, (hipStream_t)stream, a, y, mean, rstd, eps)
    if (ipt <= 1) STK_GN_FWD_FLAT(1);
    else if (ipt <= 2) STK_GN_FWD_FLAT(2);
    else if (ipt <= 3) STK_GN_FWD_FLAT(3);
    else STK_GN_FWD_FLAT(4);
#undef STK_GN_FWD_FLAT
  } else if (vec)
    hipLaunchKernelGGL((gn_fwd_kernel<4>), dim3(N * G), dim3(256), 0, (hipStream_t)stream, a, y, mean, rstd, eps);
  else
    hipLaunchKernelGGL((gn_fwd_kernel<1>), dim3(N * G), dim3(256), 0, (hipStream_t)stream, a, y, mean, rstd, eps);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

/* = stk_gn_fwd_f32 (y may be NULL when no consumer reads the fp32 copy) + stk_gn_bound_f32 (rec) +
 * stk_split_planes_f32 (planes), in one pass over x where the shape allows (gn_pl_fused_ok), else as those three. */
static int gn_fwd_pl_impl(const float* x1, int C1, const float* x2, int C2, const float* gamma, const float* beta, float* y,
                          void* planes, float* rec, float* mean, float* rstd, int N, int HW, int G, float eps, int act,
                          float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* ws, void* stream,
                          float* xmax1, float* xmax2);

int stk_gn_fwd_pl_f32(const float* x1, int C1, const float* x2, int C2, const float* gamma, const float* beta, float* y,
                      void* planes, float* rec, float* mean, float* rstd, int N, int HW, int G, float eps, int act,
                      float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* ws, void* stream) {
  return gn_fwd_pl_impl(x1, C1, x2, C2, gamma, beta, y, planes, rec, mean, rstd, N, HW, G, eps, act, drop_p, seed, seed_dev, ws,
                        stream, nullptr, nullptr);
}

/* stk_gn_fwd_pl_f32 on a shape the one-pass kernel takes (stk_gn_fwd_pl_fused) that also leaves the scale records of its
 * SOURCE tensors behind: xmax1[0..256) / xmax2[0..256) receive max |x1| / max |x2| by atomic maximum (caller-zeroed). */
int stk_gn_fwd_pl_max_f32(const float* x1, int C1, const float* x2, int C2, const float* gamma, const float* beta, float* y,
                          void* planes, float* rec, float* mean, float* rstd, int N, int HW, int G, float eps, int act,
                          float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* ws,
                          float* xmax1, float* xmax2, void* stream) {
  if (!xmax1 || (C2 > 0 && !xmax2) || !(gn_pl_fused_ok(C1, C2, HW, G) || gn_pl_2k_ok(C1, C2, HW, G))) return STK_EINVAL;
  return gn_fwd_pl_impl(x1, C1, x2, C2, gamma, beta, y, planes, rec, mean, rstd, N, HW, G, eps, act, drop_p, seed, seed_dev, ws,
                        stream, xmax1, xmax2);
}

static int gn_fwd_pl_impl(const float* x1, int C1, const float* x2, int C2, const float* gamma, const float* beta, float* y,
                          void* planes, float* rec, float* mean, float* rstd, int N, int HW, int G, float eps, int act,
                          float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* ws, void* stream,
                          float* xmax1, float* xmax2) {
  const int C = C1 + C2;
  if (!x1 || !gamma || !beta || !planes || !rec || !mean || !rstd || N <= 0 || HW <= 0 || G <= 0 || C1 <= 0 || C2 < 0 ||
      C % G || (C2 > 0 && !x2) || drop_p < 0.f || drop_p >= 1.f || act < 0 || act > STK_ACT_ELU)
    return STK_EINVAL;
  const long plane_stride = (long)N * ((C + 31) / 32) * HW * 64;
  if (2 * plane_stride >= 0x7fffffffL) return STK_EUNSUPPORTED;
  if (gn_pl_2k_ok(C1, C2, HW, G) && stk_aligned16(x1) && (!x2 || stk_aligned16(x2)) && (!y || stk_aligned16(y))) {
    GnArgs a;
    a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.gamma = gamma; a.beta = beta;
    a.N = N; a.HW = HW; a.G = G; a.cpg = C / G; a.act = act; a.drop_p = drop_p; a.keep_scale = 1.f / (1.f - drop_p); a.drop_thr = stk_drop_threshold(drop_p);
    a.seed = seed; a.seed_dev = seed_dev;
    const float sq = sqrtf((float)((long)a.cpg * HW) - 1.f);
    hipStream_t s = (hipStream_t)stream;
    if ((long)a.cpg * HW <= 16384) {
      hipLaunchKernelGGL(gn_stats_kernel, dim3(N * G), dim3(256), 0, s, a, mean, rstd, eps, sq, rec);
    } else {
      if (!ws) return STK_EINVAL;
      const int Sc = HW / GN_CHUNK;
      hipLaunchKernelGGL(gn_split_stats_kernel, dim3((unsigned)((long)N * C * Sc)), dim3(256), 0, s, a, ws, Sc);
      hipLaunchKernelGGL(gn_fold_stats_kernel, dim3(N * G), dim3(256), 0, s, a, ws, Sc, mean, rstd, eps);
      hipLaunchKernelGGL(gn_bound_kernel, dim3(1), dim3(256), 0, s, gamma, beta, C, sq, a.keep_scale, rec);
    }
    STK_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_apply_pl_kernel, dim3((unsigned)((long)N * (C / 32) * (HW / AP_PIX))), dim3(256), 0, s, a, mean, rstd, rec, y,
                       static_cast<unsigned char*>(planes), plane_stride, xmax1, xmax2);
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
  if (gn_pl_fused_ok(C1, C2, HW, G)) {
    GnArgs a;
    a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.gamma = gamma; a.beta = beta;
    a.N = N; a.HW = HW; a.G = G; a.cpg = C / G; a.act = act; a.drop_p = drop_p; a.keep_scale = 1.f / (1.f - drop_p); a.drop_thr = stk_drop_threshold(drop_p);
    a.seed = seed; a.seed_dev = seed_dev;
    // HW = 1024: two workgroups of 512 threads per CU instead of one of 1024: the same 16 waves per CU, but the load phase of
    // one block overlaps the arithmetic / store phase of the other
    constexpr int tmax = 512;
    int T = HW * 4 < 1024 ? HW * 4 : 1024;
    if (HW * 4 > 1024 && tmax < T) T = tmax;
    const int items = HW * 4, passes = items / T;
    const float sq = sqrtf((float)((long)a.cpg * HW) - 1.f);
    const dim3 grid((unsigned)(N * (C / 32)));
#define STK_GN_PL(P)                                                                                               \
  hipLaunchKernelGGL((gn_fwd_pl_kernel<P>), grid, dim3(T), 0, (hipStream_t)stream, a, y, static_cast<unsigned char*>(planes), \
                     plane_stride, rec, mean, rstd, eps, sq, xmax1, xmax2)
    if (passes == 1) STK_GN_PL(1); else if (passes == 2) STK_GN_PL(2); else if (passes == 4) STK_GN_PL(4); else if (passes == 8) STK_GN_PL(8); else return STK_EUNSUPPORTED;
#undef STK_GN_PL
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
  // the unfused route goes through the fp32 copy and leaves no |x| records: a caller that relied on stk_gn_fwd_pl_fused()
  // for a shape whose pointers miss the 16-byte alignment of the two-kernel route gets an error, not a fault / stale records
  if (!y || xmax1 || xmax2) return STK_EINVAL;
  int rc = stk_gn_fwd_f32(x1, C1, x2, C2, gamma, beta, y, mean, rstd, N, HW, G, eps, act, drop_p, seed, seed_dev, ws, stream);
  if (rc) return rc;
  rc = stk_gn_bound_f32(gamma, beta, C, G, HW, drop_p, rec, stream);
  if (rc) return rc;
  return stk_split_planes_f32(y, N, C, HW, rec, 256, planes, stream);
}

/* 1 if stk_gn_fwd_pl_f32 takes this shape in one pass (and therefore accepts y = NULL) */
int stk_gn_fwd_pl_fused(int C1, int C2, int HW, int G) { return gn_pl_fused_ok(C1, C2, HW, G) || gn_pl_2k_ok(C1, C2, HW, G) ? 1 : 0; }

/* shapes whose backward runs on the register-resident kernel and can therefore leave the by-products behind */
static inline bool gn_bwd_flat_shape(int C, int HW, int G) {
  if (C <= 0 || G <= 0 || C % G || HW < 16 || (HW & (HW - 1))) return false;
  const long L = (long)(C / G) * HW;
  return L <= 16384 && C / G <= 512 && !gn_split_ok(HW, C / G);
}
int stk_gn_bwd_out_ok(int C1, int C2, int HW, int G) { return C1 > 0 && C2 >= 0 && gn_bwd_flat_shape(C1 + C2, HW, G) ? 1 : 0; }

static int gn_bwd_impl(const float* dy, const float* x1, int C1, const float* x2, int C2, const float* gamma,
                       const float* beta, const float* mean, const float* rstd, float* dx1, float dx1_beta, float* dx2,
                       float dx2_beta, float* dgamma, float* dbeta, float* ws, int N, int HW, int G, int act, float drop_p,
                       unsigned long long seed, const unsigned long long* seed_dev, void* stream, GnBwdOut out);

int stk_gn_bwd_f32(const float* dy, const float* x1, int C1, const float* x2, int C2, const float* gamma,
                   const float* beta, const float* mean, const float* rstd, float* dx1, float dx1_beta, float* dx2,
                   float dx2_beta, float* dgamma, float* dbeta, float* ws, int N, int HW, int G, int act, float drop_p,
                   unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
  const GnBwdOut none = {nullptr, nullptr, 0, 1.f, nullptr, nullptr, 0.f};
  return gn_bwd_impl(dy, x1, C1, x2, C2, gamma, beta, mean, rstd, dx1, dx1_beta, dx2, dx2_beta, dgamma, dbeta, ws, N, HW, G, act,
                     drop_p, seed, seed_dev, stream, none);
}

int stk_gn_bwd_out_f32(const float* dy, const float* x1, int C1, const float* x2, int C2, const float* gamma,
                       const float* beta, const float* mean, const float* rstd, float* dx1, float dx1_beta, float* dx2,
                       float dx2_beta, float* dgamma, float* dbeta, float* ws, int N, int HW, int G, int act, float drop_p,
                       unsigned long long seed, const unsigned long long* seed_dev, const float* dx1_add, float add_scale,
                       float* dx_sum, float out_scale, float* dtemb, int temb_stride, float* dx_amax, void* stream) {
  if (!dx1 || !stk_gn_bwd_out_ok(C1, C2, HW, G) || (dtemb && temb_stride < C1) || (C2 > 0 && !x2)) return STK_EINVAL;
  if (!(stk_aligned16(x1) && stk_aligned16(dy) && stk_aligned16(dx1) && (!x2 || stk_aligned16(x2)) &&
        (!dx2 || stk_aligned16(dx2)) && (!dx1_add || stk_aligned16(dx1_add))))
    return STK_EUNSUPPORTED;
  const GnBwdOut out = {dx_sum, dtemb, temb_stride, out_scale, dx_amax, dx1_add, add_scale};
  return gn_bwd_impl(dy, x1, C1, x2, C2, gamma, beta, mean, rstd, dx1, dx1_beta, dx2, dx2_beta, dgamma, dbeta, ws, N, HW, G,
                     act, drop_p, seed, seed_dev, stream, out);
}

static int gn_bwd_impl(const float* dy, const float* x1, int C1, const float* x2, int C2, const float* gamma,
                       const float* beta, const float* mean, const float* rstd, float* dx1, float dx1_beta, float* dx2,
                       float dx2_beta, float* dgamma, float* dbeta, float* ws, int N, int HW, int G, int act, float drop_p,
                       unsigned long long seed, const unsigned long long* seed_dev, void* stream, GnBwdOut out) {
  const int C = C1 + C2;
  if (!dy || !x1 || !gamma || !beta || !mean || !rstd || !ws || N <= 0 || HW <= 0 || G <= 0 || C1 <= 0 || C2 < 0 ||
      C % G || (C2 > 0 && !x2) || drop_p < 0.f || drop_p >= 1.f || act < 0 || act > STK_ACT_ELU)
    return STK_EINVAL;
  GnArgs a;
  a.x1 = x1; a.x2 = x2; a.C1 = C1; a.C2 = C2; a.gamma = gamma; a.beta = beta;
  a.N = N; a.HW = HW; a.G = G; a.cpg = C / G; a.act = act; a.drop_p = drop_p; a.keep_scale = 1.f / (1.f - drop_p); a.drop_thr = stk_drop_threshold(drop_p);
  a.seed = seed; a.seed_dev = seed_dev;
  const long L = (long)a.cpg * HW;
  int hw_log2 = 0;
  while ((1 << hw_log2) < HW) ++hw_log2;
  const bool flat = (1 << hw_log2) == HW && HW >= 16 && L <= 16384 && a.cpg <= 512 && stk_aligned16(x1) &&
                    stk_aligned16(dy) && (!x2 || stk_aligned16(x2)) && (!dx1 || stk_aligned16(dx1)) &&
                    (!dx2 || stk_aligned16(dx2));
  const bool al16 = stk_aligned16(x1) && stk_aligned16(dy) && (!x2 || stk_aligned16(x2)) && (!dx1 || stk_aligned16(dx1)) &&
                    (!dx2 || stk_aligned16(dx2));
  if (al16 && gn_split_ok(HW, a.cpg)) {
    const int Sc = HW / GN_CHUNK;
    float* part = ws + 2L * N * C;                       // after the [N][C][2] channel sums
    const dim3 grid((unsigned)((long)N * C * Sc));
    hipLaunchKernelGGL(gn_split_bwd_part_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, dy, mean, rstd, part, Sc);
    STK_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_split_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, dy, mean, rstd, part, Sc, dx1,
                       dx1_beta, dx2, dx2_beta, ws);
  } else if (flat) {
    const int L4 = (int)(L >> 2);
    const int tgt = 4;          // float4 per thread (measured best of 1..4 on the 32x32 / 16x16 layers)
    int T = 64;
    while (T < 1024 && T * tgt < L4) T <<= 1;
    const int ipt = stk_cdiv(L4, T);
#define STK_GN_FLAT(IPT)                                                                                          \
  do {                                                                                                            \
    if (out.sum || out.temb || out.amax || out.add)                                                               \
      hipLaunchKernelGGL((gn_bwd_flat_kernel<IPT, true>), dim3(N * G), dim3(T), 0, (hipStream_t)stream, a, dy, mean, rstd, dx1, \
                         dx1_beta, dx2, dx2_beta, ws, hw_log2, out);                                               \
    else                                                                                                          \
      hipLaunchKernelGGL((gn_bwd_flat_kernel<IPT, false>), dim3(N * G), dim3(T), 0, (hipStream_t)stream, a, dy, mean, rstd, dx1, \
                         dx1_beta, dx2, dx2_beta, ws, hw_log2, out);                                               \
  } while (0)
    if (ipt <= 1) STK_GN_FLAT(1);
    else if (ipt <= 2) STK_GN_FLAT(2);
    else if (ipt <= 3) STK_GN_FLAT(3);
    else STK_GN_FLAT(4);
#undef STK_GN_FLAT
  } else {
    hipLaunchKernelGGL(gn_bwd_kernel, dim3(N * G), dim3(256), 0, (hipStream_t)stream, a, dy, mean, rstd, dx1, dx1_beta,
                       dx2, dx2_beta, ws);
  }
  STK_CHECK_LAUNCH();
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3(stk_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, ws, dgamma,
                       dbeta, N, C, (const StkGnFoldDesc*)nullptr);
    STK_CHECK_LAUNCH();
  }
  return STK_OK;
}

int stk_gn_param_grad_batch(const StkGnFoldDesc* descs_dev, int count, int max_C, void* stream) {
  if (!descs_dev || count <= 0 || max_C <= 0) return STK_EINVAL;
  hipLaunchKernelGGL(gn_param_grad_kernel, dim3(stk_cdiv(max_C, 32), (unsigned)count), dim3(256), 0, (hipStream_t)stream,
                     (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0, 0, descs_dev);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

long stk_gn_ws_bytes(int N, int C, int HW, int G) {
  if (N <= 0 || C <= 0 || HW <= 0 || G <= 0 || C % G) return 0;
  long f = 2L * N * C;                                   // backward: per-(sample, channel) sums
  if (gn_split_ok(HW, C / G)) f += 2L * N * C * (HW / GN_CHUNK);     // per-chunk partials (forward and backward)
  return f * 4;
}

}  // extern "C"
