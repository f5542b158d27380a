// cores.hip -- stand-alone hunt for the co-residency divergence of round 3 (profiles/r03_side_stream_race.txt).
//
// Victim: gn_bwd_flat_kernel<1, true> of soft-truncation_amd/csrc/groupnorm.hip, loaded from a code object (so that variants
// of its ISA -- SLP on / off, hand-edited assembly -- can be swapped without rebuilding anything else), launched on stream A
// on fixed inputs.  Aggressor: a synthetic kernel of one class (matrix pipe, LDS, VALU, HBM stream, idle resident waves) on
// stream B.  The victim's result is compared bit for bit with its result on an idle chip and with a float64 host evaluation.
//
//   cores <victim.hsaco> [aggressors...]      aggressors: none mfma lds valu mem idle  (default: all)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// PODs of groupnorm.hip (kernel arguments by value; probe-only mirror)
struct GnArgs {
  const float* x1; const float* x2; int C1, C2; const float* gamma; const float* beta; int N, HW, G, cpg; int act; float drop_p;
  float keep_scale; unsigned drop_thr; unsigned long long seed; const unsigned long long* seed_dev;
};
struct GnBwdOut { float* sum; float* temb; int temb_stride; float scale; float* amax; const float* add; float add_scale; };
static_assert(sizeof(GnArgs) == 88 && sizeof(GnBwdOut) == 48, "layout of groupnorm.hip");

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void agg_mfma(float* out, int iters) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void agg_lds(float* out, int iters) {
  extern __shared__ float sm[];
  const int n = 8192;
  for (int i = threadIdx.x; i < n; i += 256) sm[i] = (float)i;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < iters; ++i) {
    const int j = (threadIdx.x * 4 + i * 68) & (n - 4);
    float4 v = *reinterpret_cast<float4*>(sm + j);
    s += v.x + v.y + v.z + v.w;
    *reinterpret_cast<float4*>(sm + ((j + 2048) & (n - 4))) = make_float4(s, v.x, v.y, v.z);
  }
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void agg_valu(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) { a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.99999f, 1e-7f); }
  if (a + c + d == 12345.678f) out[0] = a;
}

__global__ __launch_bounds__(256) void agg_pk(float* out, int iters) {       // packed-fp32 VALU neighbour
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v a = {threadIdx.x * 1e-3f, 1.f}, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int i = 0; i < iters; ++i) { a = __builtin_elementwise_fma(a, b, c); c = __builtin_elementwise_fma(c, b, a); }
  if (a[0] + c[1] == 12345.678f) out[0] = a[0];
}

__global__ __launch_bounds__(256) void agg_mem(const float4* in, float4* out, long n, int passes) {
  float4 s = make_float4(0, 0, 0, 0);
  for (int p = 0; p < passes; ++p)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) { float4 v = in[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  if (s.x == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void agg_idle(float* out, int iters) {      // resident, (almost) no activity
  extern __shared__ float sm[];
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
  if (iters < 0) out[0] = sm[0];
}

// dirty the register files / LDS of every CU with a recognisable pattern and exit (what a previous kernel leaves behind)
__global__ __launch_bounds__(256) void agg_dirty(float* out, unsigned pat) {
  extern __shared__ float sm[];
  float v[96];
  const float f = __uint_as_float(pat);
#pragma unroll
  for (int i = 0; i < 96; ++i) v[i] = f + (float)i * out[1];
  for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = f;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 96; ++i) s += v[i] * sm[(i * 67 + threadIdx.x) & 16383];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void agg_imul(float* out, int iters) {       // quarter-rate integer multiplies
  unsigned a = threadIdx.x * 2654435761u + 1, b = 40503u + threadIdx.x; unsigned long long c = 1;
  for (int i = 0; i < iters; ++i) { a = a * b + 7u; c = (unsigned long long)a * b + c; b ^= (unsigned)(c >> 32); }
  if (a + b == 12345678u) out[0] = (float)c;
}

__global__ __launch_bounds__(256) void agg_mix(float* out, int iters) {        // VOP3P mixed-precision ops with op_sel
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
  unsigned h = 0;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3\n\tv_fma_mixhi_f16 %0, %2, %1, %3\n\tv_cvt_pk_f16_f32 %0, %1, %2" : "+v"(h) : "v"(a), "v"(b), "v"(c));
    a += 1e-3f;
  }
  if (h == 12345678u) out[0] = a;
}

__global__ __launch_bounds__(256) void agg_acc(float* out, int iters) {        // accumulation registers beside the victim
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a100, %0\n\ts_nop 4\n\t"
                 "v_mfma_f32_32x32x16_f16 a[0:15], %1, %2, a[0:15]\n\ts_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a5\n\tv_accvgpr_mov_b32 a101, a100"
                 : "+v"(v) : "v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a100", "a101");
  }
  if (v == 12345.678f) out[0] = v;
}

__global__ __launch_bounds__(256) void agg_bperm(float* out, int iters) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v += __shfl_xor(v, (i & 31) + 1);
  if (v == 12345.678f) out[0] = v;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: cores victim.hsaco [aggressors]\n"); return 2; }
  const int N = 96, C = 96, HW = 64, G = 24, cpg = C / G, hw_log2 = 6;
  const long E = (long)N * C * HW;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> x(E), dy(E), pre(E), gamma(C), beta(C), mean(N * G), rstd(N * G);
  for (auto& v : x) v = nd(rng);
  for (auto& v : dy) v = 0.01f * nd(rng);
  for (auto& v : pre) v = 0.03f * nd(rng);
  for (auto& v : gamma) v = 0.1f * nd(rng);
  for (auto& v : beta) v = 0.1f * nd(rng);
  for (int n = 0; n < N; ++n)
    for (int g = 0; g < G; ++g) {
      double s = 0, s2 = 0;
      const float* p = &x[((long)n * C + g * cpg) * HW];
      for (int i = 0; i < cpg * HW; ++i) { s += p[i]; s2 += (double)p[i] * p[i]; }
      const double m = s / (cpg * HW), var = s2 / (cpg * HW) - m * m;
      mean[n * G + g] = (float)m; rstd[n * G + g] = (float)(1.0 / std::sqrt(var + 1e-6));
    }
  // float64 host evaluation of dx1 = pre + rstd (du gamma - m1 - xhat m2)
  std::vector<double> want(E);
  std::vector<double> m1r(N * G);
  for (int n = 0; n < N; ++n)
    for (int g = 0; g < G; ++g) {
      const long base = ((long)n * C + g * cpg) * HW;
      const double m = mean[n * G + g], r = rstd[n * G + g];
      std::vector<double> du(cpg * HW), xh(cpg * HW);
      double g0 = 0, g1 = 0;
      for (int i = 0; i < cpg * HW; ++i) {
        const int c = g * cpg + i / HW;
        xh[i] = (x[base + i] - m) * r;
        const double u = gamma[c] * xh[i] + beta[c], sg = 1.0 / (1.0 + std::exp(-u));
        du[i] = dy[base + i] * (sg * (1.0 + u * (1.0 - sg)));
        g0 += gamma[c] * du[i]; g1 += gamma[c] * du[i] * xh[i];
      }
      const double m1 = g0 / (cpg * HW), m2 = g1 / (cpg * HW);
      m1r[n * G + g] = r * m1;
      for (int i = 0; i < cpg * HW; ++i) want[base + i] = pre[base + i] + r * (du[i] * gamma[g * cpg + i / HW] - m1 - xh[i] * m2);
    }

  const int K = 24;                      // victim launches per trial, each into its own dx buffer
  float *d_x, *d_dy, *d_pre, *d_gamma, *d_beta, *d_mean, *d_rstd, *d_dx, *d_ws, *d_sum, *d_amax, *d_out;
  CK(hipMalloc(&d_x, E * 4)); CK(hipMalloc(&d_dy, E * 4)); CK(hipMalloc(&d_pre, E * 4)); CK(hipMalloc(&d_dx, E * 4 * K));
  CK(hipMalloc(&d_gamma, C * 4)); CK(hipMalloc(&d_beta, C * 4)); CK(hipMalloc(&d_mean, N * G * 4)); CK(hipMalloc(&d_rstd, N * G * 4));
  CK(hipMalloc(&d_ws, 2L * N * C * 4)); CK(hipMalloc(&d_sum, 2L * N * C * 4)); CK(hipMalloc(&d_amax, 256 * 4)); CK(hipMalloc(&d_out, 4096));
  CK(hipMemset(d_out, 0, 4096));
  CK(hipMemcpy(d_x, x.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_dy, dy.data(), E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_pre, pre.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_gamma, gamma.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_beta, beta.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_mean, mean.data(), N * G * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rstd, rstd.data(), N * G * 4, hipMemcpyHostToDevice));
  const long big = 1L << 28;             // 256 MiB for the streaming aggressor
  float4 *d_big, *d_big_out;
  CK(hipMalloc(&d_big, big)); CK(hipMalloc(&d_big_out, 64)); CK(hipMemset(d_big, 0, big));

  hipModule_t mod;
  CK(hipModuleLoad(&mod, argv[1]));
  hipFunction_t fn;
  const char* kname = getenv("CORES_KERNEL") ? getenv("CORES_KERNEL")
      : "_ZN12_GLOBAL__N_118gn_bwd_flat_kernelILi1ELb1EEEvNS_6GnArgsEPKfS3_S3_PffS4_fS4_iNS_8GnBwdOutE";
  CK(hipModuleGetFunction(&fn, mod, kname));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));

  GnArgs a;
  memset(&a, 0, sizeof a);
  a.x1 = d_x; a.x2 = nullptr; a.C1 = C; a.C2 = 0; a.gamma = d_gamma; a.beta = d_beta; a.N = N; a.HW = HW; a.G = G; a.cpg = cpg;
  a.act = 1; a.drop_p = 0.f; a.keep_scale = 1.f; a.drop_thr = 0; a.seed = 0; a.seed_dev = nullptr;
  GnBwdOut o;
  memset(&o, 0, sizeof o);
  o.sum = d_sum; o.temb = nullptr; o.temb_stride = 0; o.scale = 1.f; o.amax = d_amax; o.add = nullptr; o.add_scale = 0.f;
  float beta1 = 1.f, beta2 = 0.f;
  float* dx2 = nullptr;
  int hl = hw_log2;
  auto victim = [&](int k, hipStream_t s) {
    float* dx1 = d_dx + (long)k * E;
    CK(hipMemcpyAsync(dx1, d_pre, E * 4, hipMemcpyDeviceToDevice, s));
    void* args[] = {&a, &d_dy, &d_mean, &d_rstd, &dx1, &beta1, &dx2, &beta2, &d_ws, &hl, &o};
    CK(hipModuleLaunchKernel(fn, N * G, 1, 1, 64, 1, 1, 0, s, args, nullptr));
  };
  std::vector<float> base(E), got(E * K);
  // baseline on an idle chip
  victim(0, sa);
  CK(hipStreamSynchronize(sa));
  CK(hipMemcpy(base.data(), d_dx, E * 4, hipMemcpyDeviceToHost));
  double werr = 0; long wbad = 0;
  for (long i = 0; i < E; ++i) { const double d = std::fabs(base[i] - want[i]); if (d > werr) werr = d; if (d > 1e-6) ++wbad; }
  printf("victim %s\nidle chip vs float64: max |err| %.3e, elements off by > 1e-6: %ld\n", argv[1], werr, wbad);

  // real aggressor: the library's own weight gradient (fp32 operands, split inside the kernel) of a 192 -> 192 3x3 layer at 8x8
  typedef long (*wsb_t)(int, int, int, int, int, int, int, int);
  typedef int (*wg_t)(const float*, int, const float*, int, const float*, float*, int, float, float*, long, int, int, int, int, int, int, int, int, int, int, void*);
  wg_t wgrad = nullptr; long wg_ws_bytes = 0; float *w_x = nullptr, *w_dy = nullptr, *w_dw = nullptr, *w_ws = nullptr;
  const int WC = getenv("CORES_WC") ? atoi(getenv("CORES_WC")) : 192, WH = getenv("CORES_WH") ? atoi(getenv("CORES_WH")) : 8;
  if (getenv("CORES_LIB")) {
    void* h = dlopen(getenv("CORES_LIB"), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    wgrad = (wg_t)dlsym(h, "stk_conv2d_wgrad_f32");
    wg_ws_bytes = ((wsb_t)dlsym(h, "stk_conv2d_wgrad_ws_bytes"))(WC, 0, N, WC, WH, WH, 3, 3);
    const long we = (long)N * WC * WH * WH;
    std::vector<float> hx(we), hd(we);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hd) v = 0.01f * nd(rng);
    CK(hipMalloc(&w_x, we * 4)); CK(hipMalloc(&w_dy, we * 4)); CK(hipMalloc(&w_dw, 9L * WC * WC * 4)); CK(hipMalloc(&w_ws, wg_ws_bytes + 256));
    CK(hipMemcpy(w_x, hx.data(), we * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w_dy, hd.data(), we * 4, hipMemcpyHostToDevice));
    CK(hipMemset(w_dw, 0, 9L * WC * WC * 4));
    printf("library aggressor: %s, wgrad %d->%d @ %dx%d, ws %ld bytes\n", getenv("CORES_LIB"), WC, WC, WH, WH, wg_ws_bytes);
  }
  std::vector<std::string> aggs;
  for (int i = 2; i < argc; ++i) aggs.push_back(argv[i]);
  if (aggs.empty()) aggs = {"none", "idle", "idle_lds", "valu", "pk", "mem", "lds", "mfma", "dirty"};
  const int iters_scale = getenv("CORES_ITERS") ? atoi(getenv("CORES_ITERS")) : 1;
  const int trials = getenv("CORES_TRIALS") ? atoi(getenv("CORES_TRIALS")) : 6;
  for (const auto& ag : aggs) {
    long launches = 0, bad_launches = 0, bad_elems = 0, bad_vs_want = 0, lane_hist[64] = {0}, comp_hist[4] = {0}, m1_like = 0;
    for (int t = 0; t < trials; ++t) {
      // aggressor on B: ~1 workgroup per CU and SIMD (grid 512 x 256 threads), long enough to cover the K victim launches
      if (ag == "mfma") hipLaunchKernelGGL(agg_mfma, dim3(512), dim3(256), 0, sb, d_out, 60000 * iters_scale);
      else if (ag == "lds") hipLaunchKernelGGL(agg_lds, dim3(512), dim3(256), 32768, sb, d_out, 400000 * iters_scale);
      else if (ag == "valu") hipLaunchKernelGGL(agg_valu, dim3(512), dim3(256), 0, sb, d_out, 1500000 * iters_scale);
      else if (ag == "pk") hipLaunchKernelGGL(agg_pk, dim3(512), dim3(256), 0, sb, d_out, 1500000 * iters_scale);
      else if (ag == "mem") hipLaunchKernelGGL(agg_mem, dim3(1024), dim3(256), 0, sb, d_big, d_big_out, big / 16, 12 * iters_scale);
      else if (ag == "idle") hipLaunchKernelGGL(agg_idle, dim3(512), dim3(256), 0, sb, d_out, 1500 * iters_scale);
      else if (ag == "idle_lds") hipLaunchKernelGGL(agg_idle, dim3(512), dim3(256), 61440, sb, d_out, 1500 * iters_scale);
      else if (ag == "dirty") {
        for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(agg_dirty, dim3(2048), dim3(256), 65536, sb, d_out, 0x7fc00000u + r);
      } else if (ag == "wgrad") {
        if (!wgrad) { fprintf(stderr, "wgrad needs CORES_LIB\n"); return 2; }
        for (int r = 0; r < 60 * iters_scale; ++r) {
          int rc = wgrad(w_x, WC, nullptr, 0, w_dy, w_dw, 0, 1.f, w_ws, wg_ws_bytes, N, WH, WH, WC, WH, WH, 3, 3, 1, 1, sb);
          if (rc) { fprintf(stderr, "wgrad rc %d\n", rc); return 1; }
        }
      } else if (ag == "wgrad_f32") {           // the same layer on the f32-input MFMA kernels (ws = NULL)
        for (int r = 0; r < 20 * iters_scale; ++r) wgrad(w_x, WC, nullptr, 0, w_dy, w_dw, 0, 1.f, nullptr, 0, N, WH, WH, WC, WH, WH, 3, 3, 1, 1, sb);
      } else if (ag == "amax") {
        typedef int (*am_t)(const float*, long, float*, void*);
        static am_t am = (am_t)dlsym(dlopen(getenv("CORES_LIB"), RTLD_NOW), "stk_amax_partial_f32");
        for (int r = 0; r < 400 * iters_scale; ++r) am(w_x, (long)N * WC * WH * WH, w_ws, sb);
      } else if (ag == "silu") {
        typedef int (*si_t)(const float*, float*, long, void*);
        static si_t si = (si_t)dlsym(dlopen(getenv("CORES_LIB"), RTLD_NOW), "stk_silu_fwd_f32");
        for (int r = 0; r < 400 * iters_scale; ++r) si(w_x, w_dy, (long)N * WC * WH * WH, sb);
      } else if (ag == "fill") {
        typedef int (*fi_t)(float*, float, long, void*);
        static fi_t fi = (fi_t)dlsym(dlopen(getenv("CORES_LIB"), RTLD_NOW), "stk_fill_f32");
        for (int r = 0; r < 400 * iters_scale; ++r) fi(w_dy, 0.5f, (long)N * WC * WH * WH, sb);
      } else if (ag == "imul") {
        hipLaunchKernelGGL(agg_imul, dim3(512), dim3(256), 0, sb, d_out, 400000 * iters_scale);
      } else if (ag == "mix") {
        hipLaunchKernelGGL(agg_mix, dim3(512), dim3(256), 0, sb, d_out, 800000 * iters_scale);
      } else if (ag == "acc") {
        hipLaunchKernelGGL(agg_acc, dim3(512), dim3(256), 0, sb, d_out, 400000 * iters_scale);
      } else if (ag == "bperm") {
        hipLaunchKernelGGL(agg_bperm, dim3(512), dim3(256), 0, sb, d_out, 800000 * iters_scale);
      } else if (ag == "short") {              // many short kernels: waves starting and ending beside the victim
        for (int r = 0; r < 3000 * iters_scale; ++r) hipLaunchKernelGGL(agg_valu, dim3(512), dim3(256), 0, sb, d_out, 200);
      } else if (ag != "none") { fprintf(stderr, "unknown aggressor %s\n", ag.c_str()); return 2; }
      CK(hipGetLastError());
      for (int k = 0; k < K; ++k) victim(k, sa);
      CK(hipStreamSynchronize(sa));
      hipError_t busy = hipStreamQuery(sb);       // still running = the victims really ran beside it
      CK(hipStreamSynchronize(sb));
      CK(hipMemcpy(got.data(), d_dx, E * 4 * K, hipMemcpyDeviceToHost));
      for (int k = 0; k < K; ++k) {
        ++launches;
        long nb = 0;
        for (long i = 0; i < E; ++i)
          if (got[k * E + i] != base[i]) {
            ++nb;
            const long in_group = i % ((long)cpg * HW);
            ++lane_hist[in_group / 4]; ++comp_hist[in_group % 4];
            const long ng = i / ((long)cpg * HW);
            const double d = (double)got[k * E + i] - base[i];
            if (std::fabs(std::fabs(d) - std::fabs(m1r[ng])) < 2e-3 * std::fabs(m1r[ng]) + 1e-9) ++m1_like;
            if (std::fabs(got[k * E + i] - want[i]) > 1e-6) ++bad_vs_want;
          }
        if (nb) { ++bad_launches; bad_elems += nb; }
      }
      if (t == 0) printf("  [%s] aggressor still running after the victims: %s\n", ag.c_str(), busy == hipErrorNotReady ? "yes" : "NO (too short)");
    }
    printf("%-9s launches %ld, differing from the idle-chip result: %ld (elements %ld, of which wrong vs float64 %ld, |delta| = rstd*m1: %ld)",
           ag.c_str(), launches, bad_launches, bad_elems, bad_vs_want, m1_like);
    if (bad_elems) {
      printf("; lanes:");
      for (int l = 0; l < 64; ++l) if (lane_hist[l]) printf(" %d", l);
      printf("; components:");
      for (int c = 0; c < 4; ++c) if (comp_hist[c]) printf(" %d", c);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
