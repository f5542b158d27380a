// elementwise.hip -- HBM-bound element-wise kernels of libstk (gfx950).
//
// All of them are pure streaming passes, so the only rules that matter are coalescing and width:
// lanes walk consecutive addresses, 16 B per lane (float4) whenever every pointer is 16-byte
// aligned and the length is a multiple of 4 (true for every buffer the engine plans: offsets are
// 256 B aligned and H*W is a multiple of 16), with a scalar path otherwise; grids are capped at
// 8 blocks per CU and grid-stride the rest.
#include "common.h"

// A product that must be ROUNDED before it is used (torch evaluates a * x + b as two kernels): HIP's __fmul_rn is a plain
// product the compiler may contract into an fma; an empty asm on the value is an optimisation barrier it cannot see through.
__device__ __forceinline__ float rounded(float v) { asm volatile("" : "+v"(v)); return v; }

namespace {

template <int V> struct Vec;
template <> struct Vec<1> {
  float v[1];
  __device__ static Vec load(const float* p, long i) { Vec r; r.v[0] = p[i]; return r; }
  __device__ void store(float* p, long i) const { p[i] = v[0]; }
};
template <> struct Vec<4> {
  float v[4];
  __device__ static Vec load(const float* p, long i) {
    float4 t = reinterpret_cast<const float4*>(p)[i];
    Vec r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
  }
  __device__ void store(float* p, long i) const {
    reinterpret_cast<float4*>(p)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// Generic launcher: F::run<V>(i) processes vector item i (elements [V*i, V*i+V)).
template <int V, class F>
__global__ __launch_bounds__(256) void ew_kernel(long nv, F f) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) f.template run<V>(i);
}

template <class F>
int launch_ew(long n, bool vec_ok, F f, hipStream_t s) {
  if (n <= 0) return STK_OK;
  if (vec_ok && (n & 3) == 0) {
    long nv = n >> 2;
    hipLaunchKernelGGL((ew_kernel<4, F>), dim3(stk_ew_grid(nv)), dim3(256), 0, s, nv, f);
  } else {
    hipLaunchKernelGGL((ew_kernel<1, F>), dim3(stk_ew_grid(n)), dim3(256), 0, s, n, f);
  }
  STK_CHECK_LAUNCH();
  return STK_OK;
}

// ---- functors -------------------------------------------------------------------------------
// layers.get_act by code (include/stk.h STK_ACT_*); the code is uniform over a launch
__device__ __forceinline__ float act_value(int act, float u) {
  switch (act) {
    case STK_ACT_SILU: return u / (1.f + expf(-u));
    case STK_ACT_RELU: return u > 0.f ? u : 0.f;
    case STK_ACT_LRELU: return u > 0.f ? u : 0.2f * u;
    case STK_ACT_ELU: return u > 0.f ? u : expm1f(u);
    default: return u;
  }
}
__device__ __forceinline__ float act_slope(int act, float u) {
  switch (act) {
    case STK_ACT_SILU: { const float sg = 1.f / (1.f + expf(-u)); return sg * (1.f + u * (1.f - sg)); }
    case STK_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case STK_ACT_LRELU: return u > 0.f ? 1.f : 0.2f;
    case STK_ACT_ELU: return u > 0.f ? 1.f : expf(u);
    default: return 1.f;
  }
}
struct ActFwd {
  const float* x; float* y; int act;
  template <int V> __device__ void run(long i) const {
    auto a = Vec<V>::load(x, i);
#pragma unroll
    for (int j = 0; j < V; ++j) a.v[j] = act_value(act, a.v[j]);
    a.store(y, i);
  }
};
struct ActBwd {
  const float* x; const float* dy; float* dx; float beta; int act;
  template <int V> __device__ void run(long i) const {
    auto a = Vec<V>::load(x, i);
    auto d = Vec<V>::load(dy, i);
    Vec<V> o;
    if (beta != 0.f) o = Vec<V>::load(dx, i);
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = (beta != 0.f ? beta * o.v[j] : 0.f) + d.v[j] * act_slope(act, a.v[j]);
    o.store(dx, i);
  }
};
struct SiluFwd {
  const float* x; float* y;
  template <int V> __device__ void run(long i) const {
    auto a = Vec<V>::load(x, i);
#pragma unroll
    for (int j = 0; j < V; ++j) a.v[j] = a.v[j] / (1.f + expf(-a.v[j]));
    a.store(y, i);
  }
};
struct SiluBwd {
  const float* x; const float* dy; float* dx; float beta;
  template <int V> __device__ void run(long i) const {
    auto a = Vec<V>::load(x, i);
    auto d = Vec<V>::load(dy, i);
    Vec<V> o;
    if (beta != 0.f) o = Vec<V>::load(dx, i);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float sg = 1.f / (1.f + expf(-a.v[j]));
      float g = d.v[j] * (sg * (1.f + a.v[j] * (1.f - sg)));
      o.v[j] = (beta != 0.f ? beta * o.v[j] : 0.f) + g;
    }
    o.store(dx, i);
  }
};
struct Axpby {
  const float* a; float alpha; const float* b; float beta; float* out;
  template <int V> __device__ void run(long i) const {
    auto x = Vec<V>::load(a, i);
    if (b && beta != 0.f) {          // beta == 0: b is not read (it may be uninitialised memory; 0 * NaN = NaN)
      auto y = Vec<V>::load(b, i);
#pragma unroll
      for (int j = 0; j < V; ++j) x.v[j] = alpha * x.v[j] + beta * y.v[j];
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) x.v[j] = alpha * x.v[j];
    }
    x.store(out, i);
  }
};
struct AddDiv {
  const float* a; const float* b; float inv; int use_inv; float* out;
  template <int V> __device__ void run(long i) const {
    auto x = Vec<V>::load(a, i);
    auto y = Vec<V>::load(b, i);
#pragma unroll
    for (int j = 0; j < V; ++j) x.v[j] = use_inv ? (x.v[j] + y.v[j]) * inv : x.v[j] + y.v[j];
    x.store(out, i);
  }
};
struct Affine {
  const float* x; float a, b; float* out;
  template <int V> __device__ void run(long i) const {
    auto v = Vec<V>::load(x, i);
#pragma unroll
    for (int j = 0; j < V; ++j) v.v[j] = a * v.v[j] + b;
    v.store(out, i);
  }
};
struct Fill {
  float v; float* out;
  template <int V> __device__ void run(long i) const {
    Vec<V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = v;
    o.store(out, i);
  }
};
struct BiasAct {   // op/fused_bias_act_kernel.cu:25-47
  const float* x; const float* b; const float* ref; float* out;
  int step_b, size_b, code; float alpha, scale;
  template <int V> __device__ void run(long i) const {
    auto v = Vec<V>::load(x, i);
    Vec<V> r;
    if (ref) r = Vec<V>::load(ref, i);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = v.v[j];
      if (b) t += b[((i * V + j) / step_b) % size_b];
      float rr = ref ? r.v[j] : 0.f;
      float y;
      switch (code) {
        default: case 10: case 11: y = t; break;
        case 12: case 32: y = 0.f; break;
        case 30: y = (t > 0.f) ? t : t * alpha; break;
        case 31: y = (rr > 0.f) ? t : t * alpha; break;
      }
      v.v[j] = y * scale;
    }
    v.store(out, i);
  }
};
struct RowScale {   // out[n,:] = x[n,:] (*|/) s[n]
  const float* x; const float* s; float* out; long inner; int mode;
  template <int V> __device__ void run(long i) const {
    auto v = Vec<V>::load(x, i);
    const float sv = s[(i * V) / inner];   // inner % V == 0 on the vector path
#pragma unroll
    for (int j = 0; j < V; ++j) v.v[j] = mode == 0 ? v.v[j] * sv : v.v[j] / sv;
    v.store(out, i);
  }
};
struct Perturb {    // out = a[n]*x + s[n]*z  (losses.py:118-119)
  const float* x; const float* z; const float* a; const float* s; float* out; long inner;
  template <int V> __device__ void run(long i) const {
    auto xv = Vec<V>::load(x, i);
    auto zv = Vec<V>::load(z, i);
    const long n = (i * V) / inner;
    const float sv = s[n];
    if (a) {
      const float av = a[n];
#pragma unroll
      for (int j = 0; j < V; ++j) xv.v[j] = rounded(av * xv.v[j]) + rounded(sv * zv.v[j]);   // torch: mean + std * z, three roundings
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) xv.v[j] = xv.v[j] + rounded(sv * zv.v[j]);
    }
    xv.store(out, i);
  }
};
struct DropMask {
  float* mask; float p, ks; unsigned long long seed;
  template <int V> __device__ void run(long i) const {
    Vec<V> v;
#pragma unroll
    for (int j = 0; j < V; ++j) v.v[j] = stk_keep(seed, (unsigned long long)(i * V + j), stk_drop_threshold(p)) ? ks : 0.f;
    v.store(mask, i);
  }
};
struct LossBwd {    // losses.py:122-132 differentiated wrt the raw network output
  const float* net; const float* z; const float* std; const float* wgt; const float* dloss; float* dnet;
  long inner; int vp, mode; float red;
  template <int V> __device__ void run(long i) const {
    auto o = Vec<V>::load(net, i);
    auto zv = Vec<V>::load(z, i);
    const long n = (i * V) / inner;
    const float sd = std[n];
    const float c = dloss[n] * wgt[n] * red * 2.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float score = vp ? -o.v[j] / sd : o.v[j];
      float r = mode == 0 ? score * sd + zv.v[j] : score + zv.v[j] / sd;
      float gs = mode == 0 ? c * r * sd : c * r;
      o.v[j] = vp ? -gs / sd : gs;
    }
    o.store(dnet, i);
  }
};

// ---- non-trivially indexed kernels ----------------------------------------------------------------
// naive_upsample_2d / naive_downsample_2d (models/up_or_down_sampling.py:59-69).  One thread per
// output pair along x so that both the 2x2 gather (down) and the 2x2 scatter (up) move float2s.
__global__ __launch_bounds__(256) void resample_up_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          long total_in, int W, float alpha, float beta) {
  // thread i handles input pixel i -> output rows 2y, 2y+1, cols 2x, 2x+1
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_in; i += stride) {
    const long row = i / W;          // p*H + y
    const int x = (int)(i - row * W);
    const float v = alpha * in[i];
    float* o0 = out + (row * 2) * (2L * W) + 2 * x;
    float* o1 = o0 + 2L * W;
    float2 a = make_float2(v, v), b = a;
    if (beta != 0.f) {
      float2 p0 = *reinterpret_cast<float2*>(o0), p1 = *reinterpret_cast<float2*>(o1);
      a.x += beta * p0.x; a.y += beta * p0.y; b.x += beta * p1.x; b.y += beta * p1.y;
    }
    *reinterpret_cast<float2*>(o0) = a;
    *reinterpret_cast<float2*>(o1) = b;
  }
}
__global__ __launch_bounds__(256) void resample_down_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            long total_out, int W2, float alpha, float beta) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total_out; i += stride) {
    const long row = i / W2;         // p*H2 + y
    const int x = (int)(i - row * W2);
    const float* s0 = in + (row * 2) * (2L * W2) + 2 * x;
    const float2 a = *reinterpret_cast<const float2*>(s0);
    const float2 b = *reinterpret_cast<const float2*>(s0 + 2L * W2);
    const float m = ((a.x + a.y) + (b.x + b.y)) * 0.25f;
    out[i] = (beta != 0.f ? beta * out[i] : 0.f) + alpha * m;
  }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                          float* __restrict__ out, int B, int dim) {
  const int half = dim / 2;
  const int total = B * half;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / half, j = i - b * half;
    const float a = t[b] * freqs[j];
    out[(long)b * dim + j] = sinf(a);
    out[(long)b * dim + half + j] = cosf(a);
    if ((dim & 1) && j == 0) out[(long)b * dim + dim - 1] = 0.f;
  }
}

__global__ void fourier_embedding_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                         float* __restrict__ out, int B, int nf) {
  const int total = B * nf;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / nf, j = i - b * nf;
    const float p = x[b] * W[j] * 2.f * 3.14159265358979323846f;
    out[(long)b * 2 * nf + j] = sinf(p);
    out[(long)b * 2 * nf + nf + j] = cosf(p);
  }
}

// One 256-thread block per sample: loss[n] = wgt[n] * red * sum r^2 (losses.py:122-132).
__global__ __launch_bounds__(256) void sm_loss_fwd_kernel(const float* __restrict__ net, const float* __restrict__ z,
                                                          const float* __restrict__ std, const float* __restrict__ wgt,
                                                          float* __restrict__ loss, long inner, int vp, int mode,
                                                          int reduce_mean) {
  __shared__ float red[8];
  const int n = blockIdx.x;
  const float sd = std[n];
  const float* o = net + (long)n * inner;
  const float* zz = z + (long)n * inner;
  float acc[1] = {0.f};
  for (long i = threadIdx.x; i < inner; i += 256) {
    // the reference's element-wise sequence, every operation rounded on its own (losses.py:122-132; models/utils.py:160)
    const float score = vp ? -o[i] / sd : o[i];
    const float r = mode == 0 ? rounded(score * sd) + zz[i] : score + zz[i] / sd;
    acc[0] += rounded(r * r);
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) {
    const float r = reduce_mean ? acc[0] / (float)inner : 0.5f * acc[0];
    loss[n] = wgt[n] * r;
  }
}

}  // namespace

#define S(stream) ((hipStream_t)(stream))

// other floating types of the native op (AT_DISPATCH_FLOATING_TYPES_AND_HALF, op/fused_bias_act_kernel.cu:77): grid-stride,
// half computes in fp32 and rounds once
template <class T, class ACC>
__global__ __launch_bounds__(256) void bias_act_t_kernel(const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ ref,
                                                         T* __restrict__ out, long n, int step_b, int size_b, int code, ACC alpha,
                                                         ACC scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    ACC t = (ACC)x[i];
    if (b) t += (ACC)b[(i / step_b) % size_b];
    const ACC rr = ref ? (ACC)ref[i] : (ACC)0;
    ACC y;
    switch (code) {
      default: case 10: case 11: y = t; break;
      case 12: case 32: y = 0; break;
      case 30: y = (t > 0) ? t : t * alpha; break;
      case 31: y = (rr > 0) ? t : t * alpha; break;
    }
    out[i] = (T)(y * scale);
  }
}
template <class T, class ACC>
static int bias_act_t(const void* x, const void* b, const void* ref, void* out, long size_x, int step_b, int size_b, int act,
                      int grad, float alpha, float scale, void* stream) {
  if (!x || !out || size_x < 0 || (b && (step_b <= 0 || size_b <= 0))) return STK_EINVAL;
  if (size_x == 0) return STK_OK;
  hipLaunchKernelGGL((bias_act_t_kernel<T, ACC>), dim3((unsigned)stk_ew_grid(size_x)), dim3(256), 0, S(stream),
                     static_cast<const T*>(x), static_cast<const T*>(b), static_cast<const T*>(ref), static_cast<T*>(out), size_x,
                     b ? step_b : 1, b ? size_b : 1, act * 10 + grad, (ACC)alpha, (ACC)scale);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

extern "C" {

int stk_silu_fwd_f32(const float* x, float* y, long n, void* stream) {
  if (!x || !y || n < 0) return STK_EINVAL;
  return launch_ew(n, stk_aligned16(x) && stk_aligned16(y), SiluFwd{x, y}, S(stream));
}

int stk_silu_bwd_f32(const float* x, const float* dy, float* dx, float beta, long n, void* stream) {
  if (!x || !dy || !dx || n < 0) return STK_EINVAL;
  return launch_ew(n, stk_aligned16(x) && stk_aligned16(dy) && stk_aligned16(dx), SiluBwd{x, dy, dx, beta}, S(stream));
}

int stk_act_fwd_f32(const float* x, float* y, long n, int act, void* stream) {
  if (!x || !y || n < 0 || act < 0 || act > STK_ACT_ELU) return STK_EINVAL;
  if (act == STK_ACT_SILU) return stk_silu_fwd_f32(x, y, n, stream);
  return launch_ew(n, stk_aligned16(x) && stk_aligned16(y), ActFwd{x, y, act}, S(stream));
}

int stk_act_bwd_f32(const float* x, const float* dy, float* dx, float beta, long n, int act, void* stream) {
  if (!x || !dy || !dx || n < 0 || act < 0 || act > STK_ACT_ELU) return STK_EINVAL;
  if (act == STK_ACT_SILU) return stk_silu_bwd_f32(x, dy, dx, beta, n, stream);
  return launch_ew(n, stk_aligned16(x) && stk_aligned16(dy) && stk_aligned16(dx), ActBwd{x, dy, dx, beta, act}, S(stream));
}

int stk_axpby_f32(const float* a, float alpha, const float* b, float beta, float* out, long n, void* stream) {
  if (!a || !out || n < 0) return STK_EINVAL;
  return launch_ew(n, stk_aligned16(a) && stk_aligned16(out) && (!b || stk_aligned16(b)),
                   Axpby{a, alpha, b, beta, out}, S(stream));
}

int stk_add_div_f32(const float* a, const float* b, float div, float* out, long n, void* stream) {
  if (!a || !b || !out || n < 0 || div == 0.f) return STK_EINVAL;
  return launch_ew(n, stk_aligned16(a) && stk_aligned16(b) && stk_aligned16(out),
                   AddDiv{a, b, 1.f / div, div != 1.f, out}, S(stream));
}

// torch.cat along the channels, materialised (Combine('cat')); one thread per output / gradient element
__global__ __launch_bounds__(256) void concat_kernel(const float* __restrict__ a, long na, const float* __restrict__ b, long nb,
                                                     float* __restrict__ out, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / (na + nb), r = i - n * (na + nb);
    out[i] = r < na ? a[n * na + r] : b[n * nb + (r - na)];
  }
}
__global__ __launch_bounds__(256) void concat_bwd_kernel(const float* __restrict__ dout, float* __restrict__ da, float beta_a, long na,
                                                         float* __restrict__ db, float beta_b, long nb, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / (na + nb), r = i - n * (na + nb);
    const float g = dout[i];
    if (r < na) { if (da) { float* q = da + n * na + r; *q = (beta_a != 0.f ? beta_a * *q : 0.f) + g; } }
    else if (db) { float* q = db + n * nb + (r - na); *q = (beta_b != 0.f ? beta_b * *q : 0.f) + g; }
  }
}
int stk_concat_f32(const float* a, int Ca, const float* b, int Cb, float* out, int N, int HW, void* stream) {
  if (!a || !b || !out || Ca <= 0 || Cb <= 0 || N <= 0 || HW <= 0) return STK_EINVAL;
  const long total = (long)N * (Ca + Cb) * HW;
  hipLaunchKernelGGL(concat_kernel, dim3((unsigned)stk_ew_grid(total)), dim3(256), 0, S(stream), a, (long)Ca * HW, b, (long)Cb * HW, out, total);
  STK_CHECK_LAUNCH();
  return STK_OK;
}
int stk_concat_bwd_f32(const float* dout, float* da, float beta_a, int Ca, float* db, float beta_b, int Cb, int N, int HW, void* stream) {
  if (!dout || (!da && !db) || Ca <= 0 || Cb <= 0 || N <= 0 || HW <= 0) return STK_EINVAL;
  const long total = (long)N * (Ca + Cb) * HW;
  hipLaunchKernelGGL(concat_bwd_kernel, dim3((unsigned)stk_ew_grid(total)), dim3(256), 0, S(stream), dout, da, beta_a, (long)Ca * HW, db, beta_b,
                     (long)Cb * HW, total);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

// FixedFouriereProjection (models/layerspp.py:31-43): one thread per input element, five outputs CHW apart
__global__ __launch_bounds__(256) void fixed_fourier_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long total, long CHW) {
  const float PI = 3.14159274101257324f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / CHW, r = i - n * CHW;
    const float v = x[i], a1 = (v * 128.f) * PI, a2 = (v * 256.f) * PI;
    float* o = y + n * 5 * CHW + r;
    o[0] = v; o[CHW] = sinf(a1); o[2 * CHW] = cosf(a1); o[3 * CHW] = sinf(a2); o[4 * CHW] = cosf(a2);
  }
}
__global__ __launch_bounds__(256) void fixed_fourier_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                                float beta, long total, long CHW) {
  const float PI = 3.14159274101257324f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long n = i / CHW, r = i - n * CHW;
    const float v = x[i], a1 = (v * 128.f) * PI, a2 = (v * 256.f) * PI;
    const float* d = dy + n * 5 * CHW + r;
    const float g = d[0] + (128.f * PI) * (cosf(a1) * d[CHW] - sinf(a1) * d[2 * CHW]) + (256.f * PI) * (cosf(a2) * d[3 * CHW] - sinf(a2) * d[4 * CHW]);
    dx[i] = (beta != 0.f ? beta * dx[i] : 0.f) + g;
  }
}
int stk_fixed_fourier_fwd_f32(const float* x, float* y, int N, int C, int HW, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || HW <= 0) return STK_EINVAL;
  const long total = (long)N * C * HW;
  hipLaunchKernelGGL(fixed_fourier_fwd_kernel, dim3((unsigned)stk_ew_grid(total)), dim3(256), 0, S(stream), x, y, total, (long)C * HW);
  STK_CHECK_LAUNCH();
  return STK_OK;
}
int stk_fixed_fourier_bwd_f32(const float* x, const float* dy, float* dx, float beta, int N, int C, int HW, void* stream) {
  if (!x || !dy || !dx || N <= 0 || C <= 0 || HW <= 0) return STK_EINVAL;
  const long total = (long)N * C * HW;
  hipLaunchKernelGGL(fixed_fourier_bwd_kernel, dim3((unsigned)stk_ew_grid(total)), dim3(256), 0, S(stream), x, dy, dx, beta, total, (long)C * HW);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_affine_f32(const float* x, float a, float b, float* out, long n, void* stream) {
  if (!x || !out || n < 0) return STK_EINVAL;
  return launch_ew(n, stk_aligned16(x) && stk_aligned16(out), Affine{x, a, b, out}, S(stream));
}

__global__ __launch_bounds__(256) void fill_strided_kernel(float* __restrict__ out, float v, long total, long len, long stride) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) out[(i / len) * stride + i % len] = v;
}
int stk_fill_strided_f32(float* out, float v, long count, long len, long stride, void* stream) {
  if (!out || count < 0 || len < 0 || stride < len) return STK_EINVAL;
  if (count * len == 0) return STK_OK;
  hipLaunchKernelGGL(fill_strided_kernel, dim3((unsigned)stk_ew_grid(count * len)), dim3(256), 0, S(stream), out, v, count * len, len, stride);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_fill_f32(float* out, float v, long n, void* stream) {
  if (!out || n < 0) return STK_EINVAL;
  return launch_ew(n, stk_aligned16(out), Fill{v, out}, S(stream));
}

int stk_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* out, long size_x,
                           int step_b, int size_b, int act, int grad, float alpha, float scale, void* stream) {
  if (!x || !out || size_x < 0 || (b && (step_b <= 0 || size_b <= 0))) return STK_EINVAL;
  const bool v = stk_aligned16(x) && stk_aligned16(out) && (!ref || stk_aligned16(ref));
  return launch_ew(size_x, v, BiasAct{x, b, ref, out, b ? step_b : 1, b ? size_b : 1, act * 10 + grad, alpha, scale},
                   S(stream));
}

int stk_fused_bias_act_f16(const void* x, const void* b, const void* ref, void* out, long size_x, int step_b, int size_b,
                           int act, int grad, float alpha, float scale, void* stream) {
  return bias_act_t<_Float16, float>(x, b, ref, out, size_x, step_b, size_b, act, grad, alpha, scale, stream);
}
int stk_fused_bias_act_f64(const double* x, const double* b, const double* ref, double* out, long size_x, int step_b, int size_b,
                           int act, int grad, float alpha, float scale, void* stream) {
  return bias_act_t<double, double>(x, b, ref, out, size_x, step_b, size_b, act, grad, alpha, scale, stream);
}

int stk_rowscale_f32(const float* x, const float* s, float* out, int N, long inner, int mode, void* stream) {
  if (!x || !s || !out || N <= 0 || inner <= 0) return STK_EINVAL;
  const bool v = stk_aligned16(x) && stk_aligned16(out) && (inner & 3) == 0;
  return launch_ew((long)N * inner, v, RowScale{x, s, out, inner, mode}, S(stream));
}

int stk_perturb_f32(const float* x, const float* z, const float* a, const float* s, float* out, int N,
                    long inner, void* stream) {
  if (!x || !z || !s || !out || N <= 0 || inner <= 0) return STK_EINVAL;
  const bool v = stk_aligned16(x) && stk_aligned16(z) && stk_aligned16(out) && (inner & 3) == 0;
  return launch_ew((long)N * inner, v, Perturb{x, z, a, s, out, inner}, S(stream));
}

// Sample post-processing (sampling_lib.py:43-44): out[n, hw, c] = uint8(clip(255 * x[n, c, hw], 0, 255)), NCHW -> NHWC.
// One thread per pixel: C strided reads (coalesced across the lanes of a wave), C contiguous bytes written.
__global__ __launch_bounds__(256) void to_uint8_nhwc_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                            long npix, int C, long HW) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += stride) {
    const long n = i / HW, hw = i - n * HW;
    const float* p = x + n * C * HW + hw;
    unsigned char* o = out + i * C;
    for (int c = 0; c < C; ++c) {
      const float v = fminf(fmaxf(p[(long)c * HW] * 255.f, 0.f), 255.f);
      o[c] = (unsigned char)v;                 // truncation, like numpy's astype(uint8) on a clipped float
    }
  }
}

int stk_samples_to_uint8(const float* x, unsigned char* out, int N, int C, long HW, void* stream) {
  if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return STK_EINVAL;
  const long npix = (long)N * HW;
  hipLaunchKernelGGL(to_uint8_nhwc_kernel, dim3(stk_ew_grid(npix)), dim3(256), 0, S(stream), x, out, npix, C, HW);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

// Input pipeline tail on the device (datasets.py:313-324 + run_lib.py:72-75): uint8 NHWC image -> float NCHW batch,
//   v = u8 / 255 (tf.image.convert_image_dtype);  per-image random horizontal flip;  uniform dequantisation
//   v = (255 v + u) / 256, u ~ U[0,1);  scaler 2v - 1 when the data is centred.
// One thread per output pixel; the draws are pure functions of (seed, image index) / (seed, output element index).
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                                            int N, int C, int H, int W, int flip, int dequant,
                                                            int centered, unsigned long long seed) {
  const long HW = (long)H * W, npix = (long)N * HW, stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += stride) {
    const long n = i / HW, hw = i - n * HW;
    const int y = (int)(hw / W), x = (int)(hw - (long)y * W);
    const bool f = flip && stk_uniform(seed ^ 0x5DEECE66DULL, (unsigned long long)n) < 0.5f;
    const unsigned char* p = img + ((n * H + y) * W + (f ? W - 1 - x : x)) * C;
    for (int c = 0; c < C; ++c) {
      const long o = (n * C + c) * HW + hw;
      float v = (float)p[c] * (1.f / 255.f);
      if (dequant) v = (255.f * v + stk_uniform(seed, (unsigned long long)o)) * (1.f / 256.f);
      if (centered) v = v * 2.f - 1.f;
      out[o] = v;
    }
  }
}

int stk_preprocess_u8(const unsigned char* img, float* out, int N, int C, int H, int W, int flip, int dequant,
                      int centered, unsigned long long seed, void* stream) {
  if (!img || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return STK_EINVAL;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(stk_ew_grid((long)N * H * W)), dim3(256), 0, S(stream), img, out, N, C, H,
                     W, flip, dequant, centered, seed);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_dropout_mask_f32(float* mask, long n, float p, unsigned long long seed, void* stream) {
  if (!mask || n < 0 || p < 0.f || p >= 1.f) return STK_EINVAL;
  return launch_ew(n, stk_aligned16(mask), DropMask{mask, p, 1.f / (1.f - p), seed}, S(stream));
}

int stk_sm_loss_fwd_f32(const float* net, const float* z, const float* std, const float* wgt, float* loss,
                        int N, long inner, int vp, int mode, int reduce_mean, void* stream) {
  if (!net || !z || !std || !wgt || !loss || N <= 0 || inner <= 0) return STK_EINVAL;
  hipLaunchKernelGGL(sm_loss_fwd_kernel, dim3(N), dim3(256), 0, S(stream), net, z, std, wgt, loss, inner, vp, mode,
                     reduce_mean);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_sm_loss_bwd_f32(const float* net, const float* z, const float* std, const float* wgt, const float* dloss,
                        float* dnet, int N, long inner, int vp, int mode, int reduce_mean, void* stream) {
  if (!net || !z || !std || !wgt || !dloss || !dnet || N <= 0 || inner <= 0) return STK_EINVAL;
  const bool v = stk_aligned16(net) && stk_aligned16(z) && stk_aligned16(dnet) && (inner & 3) == 0;
  const float red = reduce_mean ? 1.f / (float)inner : 0.5f;
  return launch_ew((long)N * inner, v, LossBwd{net, z, std, wgt, dloss, dnet, inner, vp, mode, red}, S(stream));
}

int stk_resample_naive_f32(const float* in, float* out, long planes, int H, int W, int mode, float alpha,
                           float beta, void* stream) {
  if (!in || !out || planes <= 0 || H <= 0 || W <= 0) return STK_EINVAL;
  if (mode == 0) {
    const long total = planes * H * W;
    hipLaunchKernelGGL(resample_up_kernel, dim3(stk_ew_grid(total)), dim3(256), 0, S(stream), in, out, total, W,
                       alpha, beta);
  } else if (mode == 1) {
    if ((H & 1) || (W & 1)) return STK_EINVAL;
    const long total = planes * (H / 2) * (W / 2);
    hipLaunchKernelGGL(resample_down_kernel, dim3(stk_ew_grid(total)), dim3(256), 0, S(stream), in, out, total,
                       W / 2, alpha, beta);
  } else {
    return STK_EINVAL;
  }
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_timestep_embedding_f32(const float* t, const float* freqs, float* out, int B, int dim, void* stream) {
  if (!t || !freqs || !out || B <= 0 || dim < 2) return STK_EINVAL;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(stk_ew_grid((long)B * (dim / 2))), dim3(256), 0, S(stream), t,
                     freqs, out, B, dim);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

int stk_fourier_embedding_f32(const float* x, const float* W, float* out, int B, int nf, void* stream) {
  if (!x || !W || !out || B <= 0 || nf <= 0) return STK_EINVAL;
  hipLaunchKernelGGL(fourier_embedding_kernel, dim3(stk_ew_grid((long)B * nf)), dim3(256), 0, S(stream), x, W, out,
                     B, nf);
  STK_CHECK_LAUNCH();
  return STK_OK;
}

const char* stk_strerror(int code) {
  switch (code) {
    case STK_OK: return "ok";
    case STK_EINVAL: return "invalid argument";
    case STK_ELAUNCH: return "kernel launch failed";
    case STK_EUNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}
const char* stk_backend(void) { return "hip-gfx950"; }
int stk_version(void) { return 1; }

}  // extern "C"
