// reduce_optim.hip -- row softmax, bias/time-embedding gradients, and the optimizer-side kernels
// (sum of squares for gradient clipping, fused clip+Adam, EMA) for gfx950.  All HBM-bound.
#include "common.h"

namespace {

// ---- softmax over the last dimension (AttnBlockpp, models/layerspp.py:95-97) ----------------------
// One wave64 per row: lanes stride the row, wave shuffles do the max / sum reductions; a 256-thread
// block handles 4 rows.  cols is T = H*W <= 256 for every shipped config (<= 4 values per lane).
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          long rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + row * cols;
  float* q = y + row * cols;
  float mx = -INFINITY;
  for (int i = lane; i < cols; i += 64) mx = fmaxf(mx, p[i] * scale);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < cols; i += 64) {
    const float e = expf(p[i] * scale - mx);
    q[i] = e;
    s += e;
  }
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int i = lane; i < cols; i += 64) q[i] *= inv;
}

// dx = scale * y * (dy - sum(y*dy));  dx may alias dy.
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* dy, float* dx,
                                                          long rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = y + row * cols;
  const float* d = dy + row * cols;
  float* q = dx + row * cols;
  float s = 0.f;
  for (int i = lane; i < cols; i += 64) s += p[i] * d[i];
  s = wave_sum(s);
  for (int i = lane; i < cols; i += 64) q[i] = scale * (p[i] * (d[i] - s));
}

// ---- bias / time-embedding gradients ------------------------------------------------------------------
// rowsum[n*stride + c] = alpha * sum_hw dy[n,c,:]   -- one wave per (n,c) row of HW contiguous floats.
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ dy, float* __restrict__ out, int N,
                                                     int C, int HW, int out_stride, float alpha) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)N * C) return;
  const float* p = dy + row * HW;
  float s = 0.f;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int i = lane; i < (HW >> 2); i += 64) {
      const float4 v = p4[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int i = lane; i < HW; i += 64) s += p[i];
  }
  s = wave_sum(s);
  if (lane == 0) {
    const int n = (int)(row / C), c = (int)(row - (long)n * C);
    out[(long)n * out_stride + c] = alpha * s;
  }
}
// Long rows (large maps at small batch: 512 rows of 65536 floats at 256x256, batch 4): one 256-thread workgroup per
// row -- with one wave per row only 128 workgroups exist and each wave streams 256 KB serially.
__global__ __launch_bounds__(256) void rowsum_block_kernel(const float* __restrict__ dy, float* __restrict__ out, int N,
                                                           int C, int HW, int out_stride, float alpha) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const float4* p4 = reinterpret_cast<const float4*>(dy + row * HW);
  float s = 0.f;
  for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
    const float4 v = p4[i];
    s += (v.x + v.y) + (v.z + v.w);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int n = (int)(row / C), c = (int)(row - (long)n * C);
    out[(long)n * out_stride + c] = alpha * ((red[0] + red[1]) + (red[2] + red[3]));
  }
}
// Fused form for maps below 64x64: one 1024-thread workgroup per channel, wave w sums rows n = w, w+16, ... (two rows
// in flight per wave), writes the per-(n,c) sums when the caller wants them (time-embedding gradient) and the sixteen
// per-wave totals meet in LDS in a fixed order.  One launch instead of rowsum + colsum (the second one was ~6 us of
// pure launch latency, 144 times per training step).
// AMAX: also track max |dy| of the channel (the sums need every element anyway) and leave it in amax[c] -- a planes scale
// record of the gradient tensor (include/stk.h "Planes") with one entry per channel, zero-filled up to 256 entries, so
// the data-gradient call that follows needs no |dy| pass of its own.
// RES: the row also is the gradient of a residual branch, res = alpha * dy + rbeta * res (the skip of a ResnetBlock /
// attention block whose last convolution this is, models/layerspp.py:104,287) -- written on the way instead of by a
// separate 3-stream axpby pass.
template <bool AMAX, bool RES>
__device__ __forceinline__ float row_sum_max(const float* __restrict__ p, int HW, int lane, bool vec, float& m,
                                             float* __restrict__ res, float alpha, float rbeta) {
  float s = 0.f;
  if (vec) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    float4* r4 = reinterpret_cast<float4*>(res);
    for (int i = lane; i < (HW >> 2); i += 64) {
      const float4 v = p4[i];
      s += (v.x + v.y) + (v.z + v.w);
      if (AMAX) m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      if (RES) {
        float4 o = make_float4(alpha * v.x, alpha * v.y, alpha * v.z, alpha * v.w);
        if (rbeta != 0.f) {
          const float4 r = r4[i];
          o.x = __fmaf_rn(rbeta, r.x, o.x); o.y = __fmaf_rn(rbeta, r.y, o.y);
          o.z = __fmaf_rn(rbeta, r.z, o.z); o.w = __fmaf_rn(rbeta, r.w, o.w);
        }
        r4[i] = o;
      }
    }
  } else {
    for (int i = lane; i < HW; i += 64) {
      s += p[i];
      if (AMAX) m = fmaxf(m, fabsf(p[i]));
      if (RES) res[i] = rbeta != 0.f ? __fmaf_rn(rbeta, res[i], alpha * p[i]) : alpha * p[i];
    }
  }
  return s;
}
template <bool AMAX, bool RES = false>
__global__ __launch_bounds__(1024) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ rows,
                                                         int rows_stride, float* __restrict__ dbias, int N, int C,
                                                         int HW, float alpha, float* __restrict__ amax,
                                                         float* __restrict__ res = nullptr, float rbeta = 0.f,
                                                         float* __restrict__ dbias2 = nullptr,
                                                         float* __restrict__ amax2 = nullptr) {
  __shared__ float red[16];
  __shared__ float redm[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = blockIdx.x;
  const bool vec = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
                   (!RES || (reinterpret_cast<uintptr_t>(res) & 15) == 0);
  float tot = 0.f, mx = 0.f;
  for (int n = wv; n < N; n += 32) {
    const int n2 = n + 16;
    const long o0 = ((long)n * C + c) * HW, o1 = ((long)n2 * C + c) * HW;
    float s0 = row_sum_max<AMAX, RES>(dy + o0, HW, lane, vec, mx, RES ? res + o0 : nullptr, alpha, rbeta);
    float s1 = n2 < N ? row_sum_max<AMAX, RES>(dy + o1, HW, lane, vec, mx, RES ? res + o1 : nullptr, alpha, rbeta) : 0.f;
    s0 = alpha * wave_sum(s0);
    s1 = alpha * wave_sum(s1);
    if (rows && lane == 0) {
      rows[(long)n * rows_stride + c] = s0;
      if (n2 < N) rows[(long)n2 * rows_stride + c] = s1;
    }
    tot += s0;
    if (n2 < N) tot += s1;
  }
  if (AMAX) mx = wave_max(mx);
  if (lane == 0) { red[wv] = tot; if (AMAX) redm[wv] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q];
    if (dbias) dbias[c] += t;
    if (dbias2) dbias2[c] += t;                      // a second layer with the same output gradient (see _dual below)
    if (AMAX) {
      float m = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) m = fmaxf(m, redm[q]);
      amax[c] = m;
      for (int i = c + C; i < 256; i += C) amax[i] = 0.f;
      if (amax2) {
        amax2[c] = m;
        for (int i = c + C; i < 256; i += C) amax2[i] = 0.f;
      }
    }
  }
}
// dbias[c] += sum_n src[n*stride + c].  32 channels per 256-thread block: thread (c = t%32, part = t/32)
// sums every 8th row (32 consecutive floats per row segment -> coalesced), LDS folds the 8 parts.
__global__ __launch_bounds__(256) void colsum_acc_kernel(const float* __restrict__ src, float* __restrict__ dbias, int N,
                                                         int C, int stride) {
  __shared__ float red[256];
  const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < C)
    for (int n = part; n < N; n += 8) s += src[(long)n * stride + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q * 32 + cl];
    dbias[c] += t;
  }
}

// ---- sum of squares (deterministic two-stage) -------------------------------------------------------
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ x, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s[1] = {0.f};
  const long stride = (long)gridDim.x * 256;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 v = x4[i];
      s[0] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s[0] += x[i] * x[i];
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s[0] += x[i] * x[i];
  }
  block_sum<1>(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(256) void sumsq_stage2(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += (double)part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// ---- fused clip + Adam (losses.py:47-56 -> torch.optim.Adam single-tensor step) ---------------------
struct AdamArgs {
  float lr, b1, b2, eps, wd, step_size, bc2_sqrt, max_norm;
  int adamw;
};
__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, const AdamArgs& a, float coef) {
  float gi = g * coef;
  g = gi;
  float pi = p;
  if (a.wd != 0.f) {
    if (a.adamw) pi = pi * (1.f - a.lr * a.wd);
    else gi = gi + a.wd * pi;
  }
  const float mi = m + (gi - m) * (1.f - a.b1);
  const float vi = v * a.b2 + (1.f - a.b2) * gi * gi;
  const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
  p = pi - a.step_size * (mi / denom);
  m = mi;
  v = vi;
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, AdamArgs a,
                                                   const float* __restrict__ sumsq) {
  float coef = 1.f;
  if (sumsq && a.max_norm >= 0.f) {
    coef = a.max_norm / (sqrtf(sumsq[0]) + 1e-6f);
    coef = fminf(coef, 1.f);
  }
  const long stride = (long)gridDim.x * 256;
  const long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, a, coef);
    adam_one(pp.y, gg.y, mm.y, vv.y, a, coef);
    adam_one(pp.z, gg.z, mm.z, vv.z, a, coef);
    adam_one(pp.w, gg.w, mm.w, vv.w, a, coef);
    p4[i] = pp; g4[i] = gg; m4[i] = mm; v4[i] = vv;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    adam_one(p[i], g[i], m[i], v[i], a, coef);
}
__global__ __launch_bounds__(256) void adam_kernel_scalar(float* __restrict__ p, float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long n,
                                                          AdamArgs a, const float* __restrict__ sumsq) {
  float coef = 1.f;
  if (sumsq && a.max_norm >= 0.f) {
    coef = a.max_norm / (sqrtf(sumsq[0]) + 1e-6f);
    coef = fminf(coef, 1.f);
  }
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], a, coef);
}

// AMSGrad (optim.amsgrad, losses.py:33: torch.optim.Adam(amsgrad=True)): the denominator uses the running maximum of v
__global__ __launch_bounds__(256) void adam_amsgrad_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ vmax, long n, AdamArgs a,
                                                           const float* __restrict__ sumsq) {
  float coef = 1.f;
  if (sumsq && a.max_norm >= 0.f) coef = fminf(a.max_norm / (sqrtf(sumsq[0]) + 1e-6f), 1.f);
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float gi = g[i] * coef;
    g[i] = gi;
    float pi = p[i];
    if (a.wd != 0.f) {
      if (a.adamw) pi = pi * (1.f - a.lr * a.wd);
      else gi = gi + a.wd * pi;
    }
    const float mi = m[i] + (gi - m[i]) * (1.f - a.b1);
    const float vi = v[i] * a.b2 + (1.f - a.b2) * gi * gi;
    const float vm = fmaxf(vmax[i], vi);
    p[i] = pi - a.step_size * (mi / (sqrtf(vm) / a.bc2_sqrt + a.eps));
    m[i] = mi; v[i] = vi; vmax[i] = vm;
  }
}

// ---- EMA (models/ema.py:50-51) -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ s, const float* __restrict__ p, long n, float omd,
                                                  int vec) {
  const long stride = (long)gridDim.x * 256;
  if (vec) {
    const long n4 = n >> 2;
    float4* s4 = reinterpret_cast<float4*>(s);
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      float4 a = s4[i];
      const float4 b = p4[i];
      a.x = a.x - omd * (a.x - b.x);
      a.y = a.y - omd * (a.y - b.y);
      a.z = a.z - omd * (a.z - b.z);
      a.w = a.w - omd * (a.w - b.w);
      s4[i] = a;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s[i] = s[i] - omd * (s[i] - p[i]);
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s[i] = s[i] - omd * (s[i] - p[i]);
  }
}

}  // namespace

#define S(stream) ((hipStream_t)(stream))

extern "C" {

int stk_softmax_fwd_f32(const float* x, float* y, long rows, int cols, float scale, void* stream) {
  if (!x || !y || rows <= 0 || cols <= 0) return STK_EINVAL;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(stk_cdiv(rows, 4)), dim3(256), 0, S(stream), x, y, rows, cols, scale);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_softmax_bwd_f32(const float* y, const float* dy, float* dx, long rows, int cols, float scale, void* stream) {
  if (!y || !dy || !dx || rows <= 0 || cols <= 0) return STK_EINVAL;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(stk_cdiv(rows, 4)), dim3(256), 0, S(stream), y, dy, dx, rows, cols, scale);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

/* stk_bias_grad_f32 that also leaves a planes scale record of dy in amax[0..256): per-channel max |dy| (C <= 256 entries,
 * zeros behind them).  One pass over dy for maps below 64 x 64; larger maps take the two calls it stands for. */
int stk_bias_grad_amax_f32(const float* dy, int N, int C, int HW, float alpha, float* dtemb, int temb_stride, float* dbias,
                           float* amax, float* ws, void* stream) {
  if (!dy || !amax || N <= 0 || C <= 0 || C > 256 || HW <= 0 || (!dtemb && !dbias && !ws)) return STK_EINVAL;
  if (HW < 4096) {
    hipLaunchKernelGGL(bias_grad_kernel<true>, dim3((unsigned)C), dim3(1024), 0, S(stream), dy, dtemb, temb_stride, dbias, N,
                       C, HW, alpha, amax);
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
  if (dtemb || dbias) {
    const int rc = stk_bias_grad_f32(dy, N, C, HW, alpha, dtemb, temb_stride, dbias, ws, stream);
    if (rc) return rc;
  }
  return stk_amax_partial_f32(dy, (long)N * C * HW, amax, stream);
}

/* stk_bias_grad_amax_f32 + dres = alpha * dy + dres_beta * dres in the same pass over dy (the residual branch's gradient). */
int stk_bias_grad_amax_res_f32(const float* dy, int N, int C, int HW, float alpha, float* dtemb, int temb_stride,
                               float* dbias, float* amax, float* dres, float dres_beta, float* ws, void* stream) {
  if (!dy || !amax || !dres || N <= 0 || C <= 0 || C > 256 || HW <= 0 || (!dtemb && !dbias && !ws)) return STK_EINVAL;
  if (HW < 4096) {
    hipLaunchKernelGGL((bias_grad_kernel<true, true>), dim3((unsigned)C), dim3(1024), 0, S(stream), dy, dtemb, temb_stride,
                       dbias, N, C, HW, alpha, amax, dres, dres_beta);
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
  const int rc = stk_axpby_f32(dy, alpha, dres, dres_beta, dres, (long)N * C * HW, stream);
  if (rc) return rc;
  return stk_bias_grad_amax_f32(dy, N, C, HW, alpha, dtemb, temb_stride, dbias, amax, ws, stream);
}

/* stk_bias_grad_amax_f32 for TWO layers that share dy: dbias2 += the same sums, amax2 = the same record. */
int stk_bias_grad_amax_dual_f32(const float* dy, int N, int C, int HW, float alpha, float* dtemb, int temb_stride,
                                float* dbias, float* amax, float* dbias2, float* amax2, float* ws, void* stream) {
  if (!dy || !amax || !amax2 || N <= 0 || C <= 0 || C > 256 || HW <= 0 || HW >= 4096) return STK_EINVAL;
  hipLaunchKernelGGL((bias_grad_kernel<true, false>), dim3((unsigned)C), dim3(1024), 0, S(stream), dy, dtemb, temb_stride,
                     dbias, N, C, HW, alpha, amax, (float*)nullptr, 0.f, dbias2, amax2);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_bias_grad_f32(const float* dy, int N, int C, int HW, float alpha, float* dtemb, int temb_stride, float* dbias,
                      float* ws, void* stream) {
  if (!dy || N <= 0 || C <= 0 || HW <= 0 || (!dtemb && !ws) || (!dtemb && !dbias)) return STK_EINVAL;
  if (dbias && HW < 4096) {
    hipLaunchKernelGGL(bias_grad_kernel<false>, dim3((unsigned)C), dim3(1024), 0, S(stream), dy, dtemb, temb_stride, dbias, N,
                       C, HW, alpha, (float*)nullptr);
    STK_CHECK_LAUNCH();
    return STK_OK;
  }
  float* rows = dtemb ? dtemb : ws;
  const int stride = dtemb ? temb_stride : C;
  if (HW >= 4096 && (HW & 3) == 0 && stk_aligned16(dy))
    hipLaunchKernelGGL(rowsum_block_kernel, dim3((unsigned)((long)N * C)), dim3(256), 0, S(stream), dy, rows, N, C, HW,
                       stride, alpha);
  else
    hipLaunchKernelGGL(rowsum_kernel, dim3(stk_cdiv((long)N * C, 4)), dim3(256), 0, S(stream), dy, rows, N, C, HW, stride,
                       alpha);
  STK_CHECK_LAUNCH();
  if (dbias) {
    hipLaunchKernelGGL(colsum_acc_kernel, dim3(stk_cdiv(C, 32)), dim3(256), 0, S(stream), rows, dbias, N, C, stride);
    STK_CHECK_LAUNCH();
  }
  return STK_OK;
}

int stk_sumsq_f32(const float* x, long n, float* out, float* ws, void* stream) {
  if (!x || !out || !ws || n < 0) return STK_EINVAL;
  int nb = stk_cdiv(n > 0 ? n : 1, 256L * 16);
  if (nb > SUMSQ_BLOCKS) nb = SUMSQ_BLOCKS;
  hipLaunchKernelGGL(sumsq_stage1, dim3(nb), dim3(256), 0, S(stream), x, n, ws);
  STK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, S(stream), ws, nb, out);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_adam_f32(float* p, float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                 float weight_decay, int adamw, float bc1, float bc2, const float* sumsq, float max_norm,
                 void* stream) {
  if (!p || !g || !m || !v || n < 0 || bc1 == 0.f || bc2 <= 0.f) return STK_EINVAL;
  if (n == 0) return STK_OK;
  AdamArgs a;
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = weight_decay; a.adamw = adamw;
  a.step_size = lr / bc1;
  a.bc2_sqrt = sqrtf(bc2);
  a.max_norm = max_norm;
  const bool vec = stk_aligned16(p) && stk_aligned16(g) && stk_aligned16(m) && stk_aligned16(v);
  if (vec)
    hipLaunchKernelGGL(adam_kernel, dim3(stk_ew_grid(n >> 2 ? n >> 2 : 1)), dim3(256), 0, S(stream), p, g, m, v, n, a,
                       sumsq);
  else
    hipLaunchKernelGGL(adam_kernel_scalar, dim3(stk_ew_grid(n)), dim3(256), 0, S(stream), p, g, m, v, n, a, sumsq);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_adam_amsgrad_f32(float* p, float* g, float* m, float* v, float* vmax, long n, float lr, float b1, float b2, float eps,
                         float weight_decay, int adamw, float bc1, float bc2, const float* sumsq, float max_norm,
                         void* stream) {
  if (!p || !g || !m || !v || !vmax || n < 0 || bc1 == 0.f || bc2 <= 0.f) return STK_EINVAL;
  if (n == 0) return STK_OK;
  AdamArgs a;
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = weight_decay; a.adamw = adamw;
  a.step_size = lr / bc1;
  a.bc2_sqrt = sqrtf(bc2);
  a.max_norm = max_norm;
  hipLaunchKernelGGL(adam_amsgrad_kernel, dim3(stk_ew_grid(n)), dim3(256), 0, S(stream), p, g, m, v, vmax, n, a, sumsq);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_ema_f32(float* shadow, const float* p, long n, float one_minus_decay, void* stream) {
  if (!shadow || !p || n < 0) return STK_EINVAL;
  if (n == 0) return STK_OK;
  const int vec = stk_aligned16(shadow) && stk_aligned16(p);
  hipLaunchKernelGGL(ema_kernel, dim3(stk_ew_grid(vec ? (n >> 2 ? n >> 2 : 1) : n)), dim3(256), 0, S(stream), shadow, p,
                     n, one_minus_decay, vec);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

}  // extern "C"
