/*
 * stk.h -- C ABI of libstk: the MI355X (gfx950) kernels behind the Soft-Truncation
 * score-network training step and sampling loop.
 *
 * Every entry point is a plain-C function: raw device pointers, sizes, scalars and a
 * HIP stream (passed as void* so that the header needs no HIP include).  No entry
 * allocates, frees or synchronises; every launch goes to the given stream; the caller
 * owns all memory.  Return value: 0 on success, a negative STK_E* code otherwise
 * (stk_strerror() names it).  All tensors are fp32, contiguous, NCHW unless stated.
 *
 * Two libraries implement this header bit-for-bit at the interface level:
 *   - soft-truncation_amd/csrc  -> libstk.so      (hand-written HIP for gfx950, the product)
 *   - oracle/stk_ref.c          -> libstk_ref.so  (plain-C CPU restatement, test-only checker;
 *                                                  `stream` is ignored, pointers are host pointers)
 *
 * What each entry replaces in the reference (paths relative to /root/reference):
 *   stk_upfirdn2d_f32        op/upfirdn2d.cpp:12-19 (pybind `upfirdn2d`), op/upfirdn2d_kernel.cu:209-369
 *   stk_fused_bias_act_f32   op/fused_bias_act.cpp:11-17 (pybind `fused_bias_act`), op/fused_bias_act_kernel.cu:52-99
 *   stk_gn_*                 nn.GroupNorm + nn.SiLU (+ nn.Dropout) call sites, models/layerspp.py:256,277-278,90; models/ncsnpp.py:378-422
 *   stk_conv2d_*             nn.Conv2d / NIN call sites, models/layerspp.py:273-282, models/layers.py:100-124,546-555
 *   stk_gemm_f32             NIN / Linear / attention einsums, models/layerspp.py:95-99, models/ncsnpp.py:288-292
 *   stk_softmax_*            F.softmax in AttnBlockpp, models/layerspp.py:97
 *   stk_attention_*          the einsum / softmax / einsum core of AttnBlockpp, models/layerspp.py:95-99
 *   stk_resample_naive_f32   naive_upsample_2d / naive_downsample_2d, models/up_or_down_sampling.py:59-69
 *   stk_*embedding_f32       layers.get_timestep_embedding (models/layers.py:515-529), GaussianFourierProjection (models/layerspp.py:52-54)
 *   stk_perturb_f32 / stk_sm_loss_*   losses.py:116-132
 *   stk_grad_sumsq/adam/ema  losses.py:47-56 (clip_grad_norm_ + Adam), models/ema.py:43-51
 */
#ifndef STK_H
#define STK_H

#ifdef __cplusplus
extern "C" {
#endif

#define STK_OK 0
#define STK_EINVAL (-1)    /* bad argument (null pointer, non-positive size, unsupported combination) */
#define STK_ELAUNCH (-2)   /* hipLaunchKernel / hipGetLastError reported a failure */
#define STK_EUNSUPPORTED (-3)

const char* stk_strerror(int code);
/* "hip-gfx950" for the product library, "cpu-ref" for the oracle restatement. */
const char* stk_backend(void);
int stk_version(void);

/* ------------------------------------------------------------------------------------------
 * FIR resampling: pad -> zero-upsample -> FIR -> decimate, input viewed as
 * [major, in_h, in_w, minor].  out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1.
 * Same argument order as the reference pybind (op/upfirdn2d.cpp:12-14).
 * `stk_upfirdn2d_acc_f32` computes out = beta*out + result (used by the backward pass).
 * ------------------------------------------------------------------------------------------ */
int stk_upfirdn2d_f32(const float* input, const float* kernel, float* out,
                      int major, int in_h, int in_w, int minor, int kh, int kw,
                      int up_x, int up_y, int down_x, int down_y,
                      int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
int stk_upfirdn2d_acc_f32(const float* input, const float* kernel, float* out, float beta,
                          int major, int in_h, int in_w, int minor, int kh, int kw,
                          int up_x, int up_y, int down_x, int down_y,
                          int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* The other floating types the reference's pybind functions dispatch on (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * op/upfirdn2d_kernel.cu:311, op/fused_bias_act_kernel.cu:77): same arguments; `_f16` takes IEEE half tensors (fp32
 * accumulation, one rounding), `_f64` double.  The taps / bias / ref have the tensor's type, as in the reference. */
int stk_upfirdn2d_f16(const void* input, const void* kernel, void* out, int major, int in_h, int in_w, int minor,
                      int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                      int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
int stk_upfirdn2d_f64(const double* input, const double* kernel, double* out, int major, int in_h, int in_w, int minor,
                      int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                      int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
int stk_fused_bias_act_f16(const void* x, const void* b, const void* ref, void* out,
                           long size_x, int step_b, int size_b, int act, int grad,
                           float alpha, float scale, void* stream);
int stk_fused_bias_act_f64(const double* x, const double* b, const double* ref, double* out,
                           long size_x, int step_b, int size_b, int act, int grad,
                           float alpha, float scale, void* stream);

/* out[i] = act(x[i] + b[(i/step_b) % size_b]) * scale;  act*10+grad: 10/11 linear, 12 zero,
 * 30 lrelu, 31 lrelu-grad through `ref`, 32 zero (fused_bias_act_kernel.cu:36-47).
 * b == NULL: no bias; ref == NULL: reference value 0. */
int stk_fused_bias_act_f32(const float* x, const float* b, const float* ref, float* out,
                           long size_x, int step_b, int size_b, int act, int grad,
                           float alpha, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (+SiLU) (+dropout).  The input is the channel-concat of x1 [N,C1,HW] and
 * x2 [N,C2,HW] (x2 NULL / C2 = 0 for a single source); C = C1 + C2, G groups, biased variance.
 *   u = gamma_c * (x - mean_ng) * rstd_ng + beta_c ;  y = act ? u*sigmoid(u) : u ;
 *   drop_p > 0:  y = keep(seed, flat index) ? y / (1 - drop_p) : 0      (stk_rng below)
 * The effective dropout seed is seed + (seed_dev ? *seed_dev : 0): seed_dev lets a captured
 * hipGraph draw a fresh mask on every replay (the by-value seed is frozen at capture time).
 * mean/rstd [N*G] are written by fwd and read by bwd.
 * bwd: dx = dx_beta*dx + grad wrt x (split into dx1/dx2 like the input), dgamma/dbeta are
 * accumulated (+=).
 * ws: stk_gn_ws_bytes(N, C, HW, G) bytes of scratch (>= 2*N*C floats).  fwd uses it only for groups too large for
 * one workgroup (it may be NULL otherwise; with NULL such groups take the one-workgroup-per-group kernel).
 * ------------------------------------------------------------------------------------------ */
int stk_gn_fwd_f32(const float* x1, int C1, const float* x2, int C2,
                   const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                   int N, int HW, int G, float eps, int act, float drop_p,
                   unsigned long long seed, const unsigned long long* seed_dev, float* ws, void* stream);
long stk_gn_ws_bytes(int N, int C, int HW, int G);
int stk_gn_bwd_f32(const float* dy, const float* x1, int C1, const float* x2, int C2,
                   const float* gamma, const float* beta, const float* mean, const float* rstd,
                   float* dx1, float dx1_beta, float* dx2, float dx2_beta,
                   float* dgamma, float* dbeta, float* ws,
                   int N, int HW, int G, int act, float drop_p,
                   unsigned long long seed, const unsigned long long* seed_dev, void* stream);
/* stk_gn_bwd_f32 that (a) adds a second gradient branch into dx1 on the way and (b) leaves behind what the backward of the
 * convolution that PRODUCED x1 would otherwise take one more pass over dx1 for -- when this call is the last writer of
 * dx1, dx1 is that convolution's output gradient (ResnetBlockBigGANpp: Conv_0 -> + temb -> GroupNorm_1; block output ->
 * next block's GroupNorm_0, models/layerspp.py:256-287).
 *   dx1_add [N, C1, HW] (may be NULL): dx1 = dx1_beta * dx1 + (GroupNorm gradient) + add_scale * dx1_add -- the identity
 *           skip of the block this layer opens, out = (x + h) / sqrt 2: add = d(out), add_scale = 1 / sqrt 2.
 * By-products, all computed from the FINAL dx1 values, each may be NULL:
 *   dx_sum  [N][C1][2]: {s, s}, s = out_scale * sum_hw dx1[n,c,:] -- the partial-sum format of stk_gn_param_grad_batch, so
 *           up to two bias gradients (a convolution's and its shortcut peer's) are one more entry of that fold;
 *   dtemb   [n*temb_stride + c] = out_scale * sum_hw dx1[n,c,:]  (the time-embedding projection's gradient, written);
 *   dx_amax [256]: a planes scale record of dx1 by atomic maximum -- the caller ZEROES it before the launch
 *           (stk_fill_strided_f32), max(dx_amax[0..256)) = max |dx1| afterwards.
 * Shapes: stk_gn_bwd_out_ok (the register-resident backward: H*W a power of two >= 16, groups of at most 16384
 * elements); 16-byte aligned tensors. */
int stk_gn_bwd_out_ok(int C1, int C2, int HW, int G);
int stk_gn_bwd_out_f32(const float* dy, const float* x1, int C1, const float* x2, int C2,
                       const float* gamma, const float* beta, const float* mean, const float* rstd,
                       float* dx1, float dx1_beta, float* dx2, float dx2_beta,
                       float* dgamma, float* dbeta, float* ws, int N, int HW, int G, int act, float drop_p,
                       unsigned long long seed, const unsigned long long* seed_dev,
                       const float* dx1_add, float add_scale,
                       float* dx_sum, float out_scale, float* dtemb, int temb_stride, float* dx_amax, void* stream);

/* The affine-parameter gradients of MANY GroupNorm layers in one launch.  stk_gn_bwd_f32 with dgamma == dbeta == NULL
 * leaves its per-(sample, channel) sums in ws[0 .. 2*N*C) ([n][c]{sum du, sum du*xhat}) and skips its own fold; this
 * entry then does, for every descriptor, dgamma[c] += sum_n part[n][c][1], dbeta[c] += sum_n part[n][c][0] (same
 * summation order as the per-layer fold, so results are bit-identical).  A training step has ~95 GroupNorm layers:
 * one launch per backward segment instead of one 7 us launch per layer.  descs_dev: device array (host array for the
 * checker); max_C >= every descriptor's C. */
typedef struct StkGnFoldDesc {
  const float* part; float* dgamma; float* dbeta; int N, C;
} StkGnFoldDesc;
int stk_gn_param_grad_batch(const StkGnFoldDesc* descs_dev, int count, int max_C, void* stream);


/* ------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on the matrix cores.  Input = concat(x1[N,C1,H,W], x2[N,C2,H,W]).
 * w_layout 0: w[Cout][Cin][KH][KW] (nn.Conv2d);  w_layout 1: w[Cin][Cout] (NIN, KH=KW=1).
 * Any input coordinate outside [0,H)x[0,W) reads as zero, so OH/OW together with `pad`
 * (top/left) also express asymmetric padding.
 *   fwd:   y = (conv(x,w) + bias[co] + temb[n*temb_stride + co] + res[n,co,oy,ox]) / out_div
 *          (bias, temb, res may be NULL; out_div = 1 disables it; like torch's tensor / python-scalar the
 *          division is evaluated as a multiplication by the f32 reciprocal 1.f/out_div)
 *   dgrad: dx{1,2} = beta{1,2}*dx{1,2} + alpha * conv_transpose(dy, w)   (dx split by channel)
 *   wgrad: dw += alpha * sum_{n,oy,ox} dy * x        (same layout as w)
 *          ws: scratch of stk_conv2d_wgrad_ws_bytes(...) bytes for the split-K partial slabs.
 * fwd / dgrad scratch: with ws of at least stk_conv2d_{fwd,dgrad}_ws_bytes(...) bytes (0 = the shape does not
 * qualify) a 3x3 / stride-1 / pad-1 or 1x1 layer runs on the fp16 matrix pipe: each operand tensor is scaled by a
 * power of two (from its |x| maximum, computed inside the call) and every fp32 value split into two fp16 terms;
 * the three significant partial products are accumulated in fp32 and the scales undone exactly (error at the level
 * of fp32 rounding, same as the f32-input MFMA path).  ws holds the prepared weights of this call, the partial
 * maxima and the K-split slabs.  The weight gradients of these layers split both of their operands the same way.
 * ws = NULL / too small selects the f32-input MFMA path (v_mfma_f32_32x32x2_f32) for every shape.
 * ------------------------------------------------------------------------------------------ */
long stk_conv2d_fwd_ws_bytes(int C1, int C2, int N, int H, int W, int Cout, int KH, int KW,
                             int stride, int pad);
/* Which kernel family a call with full scratch takes (for profiling labels): dir 0 fwd, 1 dgrad, 2 wgrad;
 * returns 0 / 1 = f32-input MFMA with 64 / 128 tiles, (2 = the bf16 three-way split of rounds 1-5: retired) 3 = f32-input all-taps wgrad,
 * 4 = streaming kernel for a <= 4 channel side (stem, head, 3-channel pyramids), 5 = fp16 two-way split,
 * < 0 = unsupported shape. */
int stk_conv2d_variant(int dir, int C1, int C2, int N, int H, int W, int Cout, int OH, int OW,
                       int KH, int KW, int stride, int pad, int w_layout);
long stk_conv2d_dgrad_ws_bytes(int C1, int C2, int N, int H, int W, int Cout, int KH, int KW,
                               int stride, int pad);
int stk_conv2d_fwd_f32(const float* x1, int C1, const float* x2, int C2,
                       const float* w, int w_layout, const float* bias,
                       const float* temb, int temb_stride, const float* res, float out_div,
                       float* y, int N, int H, int W, int Cout, int OH, int OW,
                       int KH, int KW, int stride, int pad, void* ws, long ws_bytes, void* stream);
int stk_conv2d_dgrad_f32(const float* dy, const float* w, int w_layout,
                         float* dx1, int C1, float beta1, float* dx2, int C2, float beta2,
                         float alpha, int N, int H, int W, int Cout, int OH, int OW,
                         int KH, int KW, int stride, int pad, void* ws, long ws_bytes, void* stream);
/* Prepared weights.  The split kernel reads its weights re-laid out, scaled and split into two fp16 planes (behind a
 * 256-byte header holding their partial maxima); the calls above
 * prepare them into ws on every call.  A caller whose weights change once per optimizer step (training) or never
 * (a sampling loop: ~2000 network evaluations on fixed weights, sampling.py:365-433) prepares all layers in ONE
 * launch and hands each call its block:
 *   stk_conv2d_wp_bytes     bytes of the block of one layer and direction (dir 0 forward, 1 data gradient);
 *                           0 = this direction of this shape does not take the split kernel: pass wp = NULL
 *   stk_conv2d_wp_desc      host-side fill of one table entry (wp 256-byte aligned device memory of that size);
 *                           returns the entry's work items (> 0) or a negative error
 *   stk_conv2d_wprep_batch  one launch over a DEVICE-resident table of n entries; max_items = the largest work
 *                           item count of the table
 *   stk_conv2d_{fwd,dgrad}_wp_f32   as stk_conv2d_{fwd,dgrad}_f32 with wp = the prepared block (results are
 *                           bit-identical to the wp = NULL call); ws is still needed for the K-split slabs.
 * The caller owns coherence: a block is valid until the layer's weights change. */
typedef struct StkWprepDesc {
  const float* w; void* wp; long sm, sk; int M, Kc, Mpad, taps, flip, reserved;
} StkWprepDesc;
long stk_conv2d_wp_bytes(int dir, int C1, int C2, int N, int H, int W, int Cout, int KH, int KW,
                         int stride, int pad);
long stk_conv2d_wp_desc(int dir, const float* w, int w_layout, int Cin, int Cout, int KH, int KW,
                        void* wp, StkWprepDesc* out);
int stk_conv2d_wprep_batch(const StkWprepDesc* descs_dev, int n, long max_items, void* stream);
int stk_conv2d_fwd_wp_f32(const float* x1, int C1, const float* x2, int C2,
                          const float* w, int w_layout, const float* bias,
                          const float* temb, int temb_stride, const float* res, float out_div,
                          float* y, int N, int H, int W, int Cout, int OH, int OW,
                          int KH, int KW, int stride, int pad,
                          const void* wp, float* amax, void* ws, long ws_bytes, void* stream);
int stk_conv2d_dgrad_wp_f32(const float* dy, const float* w, int w_layout,
                            float* dx1, int C1, float beta1, float* dx2, int C2, float beta2,
                            float alpha, int N, int H, int W, int Cout, int OH, int OW,
                            int KH, int KW, int stride, int pad,
                            const void* wp, float* amax, void* ws, long ws_bytes, void* stream);
/* amax (may be NULL): a caller-owned buffer of 768 floats per layer and forward/backward pair.  The split kernels
 * scale each operand tensor by its |x| maximum; the forward call leaves the partial maxima of x1 / x2 in
 * amax[0..511], the data-gradient call those of dy in amax[512..767], and stk_conv2d_wgrad_amax_f32 reuses them
 * (have: bit 0 = the x part is valid, bit 1 = the dy part) instead of repeating the passes.  Only calls that took
 * the split kernel (stk_conv2d_variant == 5) write their part. */
int stk_conv2d_wgrad_amax_f32(const float* x1, int C1, const float* x2, int C2, const float* dy,
                              float* dw, int w_layout, float alpha, float* ws, long ws_bytes,
                              int N, int H, int W, int Cout, int OH, int OW,
                              int KH, int KW, int stride, int pad,
                              const float* amax, int have, void* stream);
long stk_conv2d_wgrad_ws_bytes(int C1, int C2, int N, int Cout, int OH, int OW, int KH, int KW);
int stk_conv2d_wgrad_f32(const float* x1, int C1, const float* x2, int C2, const float* dy,
                         float* dw, int w_layout, float alpha, float* ws, long ws_bytes,
                         int N, int H, int W, int Cout, int OH, int OW,
                         int KH, int KW, int stride, int pad, void* stream);
/* ------------------------------------------------------------------------------------------
 * Planes: an activation tensor pre-split for the fp16 two-way-split kernels (csrc/conv_pl.h).
 *   planes(N, C, HW) = P[split][n][c / 32][pixel][c % 32] IEEE binary16 of  s * x[n, c, pixel],
 *   split 0 = round-to-nearest(s x), split 1 = round-to-nearest(s x - split 0); C padded to a multiple of 32 with
 *   zeros; s = the power of two that puts max(amax[0..namax)) into [2^13, 2^14) (1 if that maximum is 0).
 * `amax` is the tensor's scale record: non-negative floats whose maximum bounds |x| -- the 256 partial maxima of
 * stk_amax_partial_f32, or an a-priori bound in amax[0] with zeros behind it.  Consumers read a record of 256.
 *   stk_planes_bytes          bytes of both planes
 *   stk_amax_partial_f32      part[0..256) = partial maxima of |x| (any partition)
 *   stk_split_planes_f32      fp32 NCHW -> planes
 *   stk_conv2d_pl_ok          1 if stk_conv2d_{fwd (dir 0), dgrad (dir 1)}_pl_f32 take this shape (3x3 pad 1 or 1x1,
 *                             stride 1, single-source operand in whole 32-channel blocks, split-kernel geometry)
 *   stk_conv2d_fwd_pl_f32     stk_conv2d_fwd_wp_f32 with x given as planes (+ its scale record); OH = H, OW = W,
 *                             stride 1, pad = KH / 2; wp may be NULL (weights are then prepared into ws)
 *   stk_conv2d_dgrad_pl_f32   stk_conv2d_dgrad_wp_f32 with dy given as planes
 * Results equal the fp32-input calls up to the position of the split (errors at the fp32 rounding level).
 * ------------------------------------------------------------------------------------------ */
/* Scale record of a GroupNorm (+SiLU) (+dropout) output from its parameters alone: rec[0] = (max|gamma| sqrt(L - 1) +
 * max|beta|) / (1 - drop_p) >= |y| for ANY input (L = (C / G) HW elements per group), rec[1..255] = 0. */
int stk_gn_bound_f32(const float* gamma, const float* beta, int C, int G, int HW, float drop_p, float* rec, void* stream);
/* GroupNorm forward whose output goes to plane consumers: stk_gn_fwd_f32 + stk_gn_bound_f32 (rec) +
 * stk_split_planes_f32 (planes) in one call -- and one pass over x for the shapes stk_gn_fwd_pl_fused reports (whole
 * groups per 32-channel block, H W in {16, 64, 256, 1024}); for those y may be NULL (no fp32 copy is written). */
int stk_gn_fwd_pl_f32(const float* x1, int C1, const float* x2, int C2, const float* gamma, const float* beta, float* y,
                      void* planes, float* rec, float* mean, float* rstd, int N, int HW, int G, float eps, int act,
                      float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* ws, void* stream);
int stk_gn_fwd_pl_fused(int C1, int C2, int HW, int G);
/* stk_gn_fwd_pl_f32 on a one-pass shape (stk_gn_fwd_pl_fused) that also leaves the planes scale records of its SOURCE
 * tensors behind -- for the ResnetBlock's 1x1 shortcut convolution, which reads the same tensors as fp32 operands of the
 * split kernels (stk_conv2d_fwd_rec_f32): xmax1[0..256) / xmax2[0..256) receive max |x1| / max |x2| by atomic maximum;
 * the caller zeroes them before the launch (stk_fill_strided_f32).  xmax2 may be NULL when C2 == 0. */
int stk_gn_fwd_pl_max_f32(const float* x1, int C1, const float* x2, int C2, const float* gamma, const float* beta, float* y,
                          void* planes, float* rec, float* mean, float* rstd, int N, int HW, int G, float eps, int act,
                          float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* ws,
                          float* xmax1, float* xmax2, void* stream);
long stk_planes_bytes(int N, int C, int HW);
int stk_amax_partial_f32(const float* x, long n, float* part, void* stream);
int stk_split_planes_f32(const float* x, int N, int C, int HW, const float* amax, int namax, void* planes, void* stream);
int stk_conv2d_pl_ok(int dir, int C1, int C2, int N, int H, int W, int Cout, int KH, int KW, int stride, int pad);
int stk_conv2d_fwd_pl_f32(const void* xpl, const float* xamax, int C, const float* w, int w_layout, const float* bias,
                          const float* temb, int temb_stride, const float* res, float out_div, float* y, int N, int H,
                          int W, int Cout, int KH, int KW, const void* wp, void* ws, long ws_bytes, void* stream);
/* K splits of the plane-operand forward (dir 0) / data-gradient (dir 1) call of a shape: 1 = one GEMM launch, > 1 =
 * partial slabs + a slab-sum launch (small maps), 0 = the shape does not take plane operands (diagnostic: kernel labels) */
int stk_conv2d_pl_ksplit(int dir, int C1, int C2, int N, int H, int W, int Cout, int KH, int KW);
/* map width W when that call runs on the halo-tile GEMM (one staged halo tile of the activations per channel group serves the
 * nine taps: 16 / 32 / 64-wide maps of whole 128-pixel tiles, no K split), else 0 (diagnostic: kernel labels) */
int stk_conv2d_pl_halo(int dir, int C1, int C2, int N, int H, int W, int Cout, int KH, int KW);
int stk_conv2d_dgrad_pl_f32(const void* dypl, const float* dyamax, const float* w, int w_layout, float* dx1, int C1,
                            float beta1, float* dx2, int C2, float beta2, float alpha, int N, int H, int W, int Cout,
                            int KH, int KW, const void* wp, void* ws, long ws_bytes, void* stream);

/* 3x3 / stride 1 / pad 1 weight gradient (w_layout 0) with BOTH operands as planes: dw += alpha * sum dy * x, the planes
 * being the ones the layer's forward (x) and data-gradient (dy) calls read.  H = W a power of two >= 8, channel counts
 * multiples of 32 (stk_conv2d_wgrad_pl_ok); ws: stk_conv2d_wgrad_pl_ws_bytes bytes for the K-split slabs. */
int stk_conv2d_wgrad_pl_ok(int N, int H, int W, int Cin, int Cout);
long stk_conv2d_wgrad_pl_ws_bytes(int N, int H, int W, int Cin, int Cout);
int stk_conv2d_wgrad_pl_f32(const void* xpl, const float* xrec, const void* dypl, const float* dyrec, float* dw,
                            float alpha, float* ws, long ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream);
/* the same with the number of workgroups its K split fills chosen by the caller: 0 = the default (one workgroup per CU: a weight gradient
 * launched on a side stream leaves half of every CU to the kernels of the main stream), e.g. 512 = two per CU for a launch that has the
 * chip to itself; <= 1024.  Results agree up to the summation order of the K-split slabs. */
int stk_conv2d_wgrad_pl_wgs_f32(const void* xpl, const float* xrec, const void* dypl, const float* dyrec, float* dw,
                                float alpha, float* ws, long ws_bytes, int N, int H, int W, int Cin, int Cout, int wgs, void* stream);

/* dtemb[n*temb_stride + c] = alpha * sum_hw dy[n,c,:]  (written; may be NULL);
 * dbias[c] += alpha * sum_{n,hw} dy[n,c,:]              (accumulated; may be NULL).
 * ws: >= N*C floats of scratch (only used when dtemb is NULL). */
int stk_bias_grad_f32(const float* dy, int N, int C, int HW, float alpha,
                      float* dtemb, int temb_stride, float* dbias, float* ws, void* stream);
/* The same plus a planes scale record of dy (see "Planes" above) in amax[0..256): amax[c] = max |dy[:, c, :]| for
 * c < C <= 256, zeros behind -- the bias gradient reads every element of dy anyway.  dtemb and dbias may both be NULL. */
int stk_bias_grad_amax_f32(const float* dy, int N, int C, int HW, float alpha,
                           float* dtemb, int temb_stride, float* dbias, float* amax, float* ws, void* stream);
/* The same plus the gradient of a residual branch in the same pass: dres = alpha * dy + dres_beta * dres (the skip
 * connection of a ResnetBlock / attention block whose last convolution produced y = (conv + res) / out_div,
 * models/layerspp.py:104,287; alpha = 1 / out_div).  dres [N, C, HW]; beta == 0: not read. */
int stk_bias_grad_amax_res_f32(const float* dy, int N, int C, int HW, float alpha,
                               float* dtemb, int temb_stride, float* dbias, float* amax,
                               float* dres, float dres_beta, float* ws, void* stream);

/* Two layers with the SAME output gradient up to a factor -- the last 3x3 convolution of a ResnetBlock and the 1x1
 * shortcut convolution on the block input, whose outputs are added (models/layerspp.py:283-287): d(shortcut out) =
 * d(block out) / out_div.  One pass over dy serves both: stk_bias_grad_amax_f32 + dbias2 += the same sums, amax2 = the
 * same record (maps below 64 x 64).  stk_conv2d_dgrad_rec_f32 = stk_conv2d_dgrad_wp_f32 whose amax[512..768) already
 * holds the record of ITS dy operand, so it makes no |dy| pass of its own (the factor goes into alpha). */
int stk_bias_grad_amax_dual_f32(const float* dy, int N, int C, int HW, float alpha,
                                float* dtemb, int temb_stride, float* dbias, float* amax,
                                float* dbias2, float* amax2, float* ws, void* stream);
/* stk_conv2d_fwd_wp_f32 whose amax[0..256) / amax[256..512) already hold the scale records of x1 / x2
 * (stk_gn_fwd_pl_max_f32): the call makes no |x| pass of its own. */
int stk_conv2d_fwd_rec_f32(const float* x1, int C1, const float* x2, int C2, const float* w, int w_layout,
                           const float* bias, const float* temb, int temb_stride, const float* res, float out_div,
                           float* y, int N, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad,
                           const void* wp, float* amax, void* ws, long ws_bytes, void* stream);
int stk_conv2d_dgrad_rec_f32(const float* dy, const float* w, int w_layout,
                             float* dx1, int C1, float beta1, float* dx2, int C2, float beta2,
                             float alpha, int N, int H, int W, int Cout, int OH, int OW,
                             int KH, int KW, int stride, int pad,
                             const void* wp, float* amax, void* ws, long ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched strided GEMM on the fp32 MFMA path:
 *   C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][k][n] + bias + beta * C[b][m][n]
 * with element (b,m,k) of A at A + b*sab + m*sam + k*sak (likewise B, C).
 * bias_mode 0: none; 1: bias[m]; 2: bias[n].
 * ------------------------------------------------------------------------------------------ */
int stk_gemm_f32(const float* A, long sam, long sak, long sab,
                 const float* B, long sbk, long sbn, long sbb,
                 float* C, long scm, long scn, long scb,
                 const float* bias, int bias_mode,
                 int M, int N, int K, int batch, float alpha, float beta, void* stream);

/* y[r,:] = softmax(scale * x[r,:]);   dx[r,:] = scale * y * (dy - sum(y*dy)) */
int stk_softmax_fwd_f32(const float* x, float* y, long rows, int cols, float scale, void* stream);
int stk_softmax_bwd_f32(const float* y, const float* dy, float* dx, long rows, int cols,
                        float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused attention core of AttnBlockpp (models/layerspp.py:95-99): the two einsums and the softmax between them,
 *   o[b,c,t] = sum_t' softmax_t'( scale * sum_c' q[b,c',t] k[b,c',t'] ) v[b,c,t'],      scale = C^-0.5,
 * on q, k, v, o [B, C, T] (NCHW with T = H*W), forward and backward, with no [B, T, T] matrix in memory.
 *   lse   [B, T]   out (forward) / in (backward): log sum_t' exp(scale * s[t, t'])
 *   rec   4 x 256 floats owned by the caller: the scale records (partial |x| maxima, see "Planes") of q, k, v written
 *         by the forward and of d_o written by the backward; the backward reads the forward's three
 *   delta [B, T]   scratch of the backward: sum_t' p[t, t'] dp[t, t']
 *   dq / dk / dv = beta * dq / dk / dv + gradient  (beta == 0: not read)
 *   qkv_bstride / grad_bstride: floats between consecutive images of q, k, v / of dq, dk, dv (C*T for separate
 *         tensors; 3*C*T when the three are the channel slices of one [B, 3C, T] tensor, as the engine's stacked
 *         q / k / v projection produces them); o and d_o are contiguous [B, C, T]
 * stk_attention_ok: 1 if the fused kernels take the shape (C % 32 == 0, 32 <= C <= 256, T % 4 == 0, T <= 256); other
 * shapes return STK_EUNSUPPORTED and go through stk_gemm_f32 / stk_softmax_*.
 * ------------------------------------------------------------------------------------------ */
int stk_attention_ok(int B, int C, int T);
int stk_attention_fwd_f32(const float* q, const float* k, const float* v, long qkv_bstride, float* o, float* lse, float* rec,
                          int B, int C, int T, float scale, void* stream);
int stk_attention_bwd_f32(const float* q, const float* k, const float* v, long qkv_bstride, const float* d_o, const float* lse,
                          float* rec, float* delta, float* dq, float beta_q, float* dk, float beta_k, float* dv, float beta_v,
                          long grad_bstride, int B, int C, int T, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise and small helpers.
 * ------------------------------------------------------------------------------------------ */
/* y = x*sigmoid(x);  dx = beta*dx + dy * sig(x)*(1 + x*(1-sig(x))) */
int stk_silu_fwd_f32(const float* x, float* y, long n, void* stream);
int stk_silu_bwd_f32(const float* x, const float* dy, float* dx, float beta, long n, void* stream);
/* The activations of layers.get_act (models/layers.py:29-41), by the `act` code every GroupNorm entry of this header takes too:
 *   STK_ACT_NONE 0 | STK_ACT_SILU 1 ('swish', nn.SiLU) | STK_ACT_RELU 2 (nn.ReLU) | STK_ACT_LRELU 3 (nn.LeakyReLU(0.2)) |
 *   STK_ACT_ELU 4 (nn.ELU(), alpha 1).   y = act(x);  dx = beta*dx + dy * act'(x)  (ReLU / LeakyReLU: act'(0) = 0 / 0.2 as torch). */
#define STK_ACT_NONE 0
#define STK_ACT_SILU 1
#define STK_ACT_RELU 2
#define STK_ACT_LRELU 3
#define STK_ACT_ELU 4
int stk_act_fwd_f32(const float* x, float* y, long n, int act, void* stream);
int stk_act_bwd_f32(const float* x, const float* dy, float* dx, float beta, long n, int act, void* stream);
/* out = alpha*a + beta*b   (b may be NULL -> treated as 0; out may alias a or b).
 * Every `beta` of this header follows the same rule: beta == 0 means the accumulated operand is NOT READ (it may be
 * uninitialised memory), never multiplied by zero. */
int stk_axpby_f32(const float* a, float alpha, const float* b, float beta, float* out, long n, void* stream);
/* out = (a + b) * (1.f/div)  -- the skip_rescale combine (x + h)/sqrt(2), models/layerspp.py:104,287 */
int stk_add_div_f32(const float* a, const float* b, float div, float* out, long n, void* stream);
/* torch.cat([a, b], dim=1) materialised (Combine(method='cat'), models/layerspp.py:57-72): out [N, Ca + Cb, HW];
 * bwd: da = beta_a*da + dout[:, :Ca], db = beta_b*db + dout[:, Ca:]  (either may be NULL). */
int stk_concat_f32(const float* a, int Ca, const float* b, int Cb, float* out, int N, int HW, void* stream);
int stk_concat_bwd_f32(const float* dout, float* da, float beta_a, int Ca, float* db, float beta_b, int Cb, int N, int HW, void* stream);
/* FixedFouriereProjection (models/layerspp.py:31-43; config.model.fourier_feature):
 *   y[n] = cat(x[n], sin(128 pi x[n]), cos(128 pi x[n]), sin(256 pi x[n]), cos(256 pi x[n])) along the channels: x [N, C, HW] ->
 *   y [N, 5 C, HW]; the arguments are formed as torch forms them (fl(fl(128 x) * fl(pi))).
 *   bwd: dx = beta*dx + dy0 + 128 pi (cos1 dy1 - sin1 dy2) + 256 pi (cos2 dy3 - sin2 dy4). */
int stk_fixed_fourier_fwd_f32(const float* x, float* y, int N, int C, int HW, void* stream);
int stk_fixed_fourier_bwd_f32(const float* x, const float* dy, float* dx, float beta, int N, int C, int HW, void* stream);
/* out = a*x + b */
int stk_affine_f32(const float* x, float a, float b, float* out, long n, void* stream);
/* out[0..n) = v */
int stk_fill_f32(float* out, float v, long n, void* stream);
/* out[i*stride + j] = v for i < count, j < len  (the engine zeroes the |dy| thirds of all its convolutions' amax buffers
 * -- the atomic-maximum records of stk_gn_bwd_out_f32 -- with one launch) */
int stk_fill_strided_f32(float* out, float v, long count, long len, long stride, void* stream);
/* mode 0: out[p,2y+i,2x+j] = alpha*in[p,y,x] (+ beta*out)   (naive_upsample_2d, in [planes,H,W])
 * mode 1: out[p,y,x] = alpha*mean_{i,j} in[p,2y+i,2x+j] (+ beta*out) (naive_downsample_2d, in [planes,H,W]) */
int stk_resample_naive_f32(const float* in, float* out, long planes, int H, int W, int mode,
                           float alpha, float beta, void* stream);
/* out[n, :] = mode==0 ? x[n,:]*s[n] : x[n,:]/s[n]   (scale_by_sigma, models/ncsnpp.py:428-430) */
int stk_rowscale_f32(const float* x, const float* s, float* out, int N, long inner, int mode, void* stream);
/* positional embedding: out[b, j] = sin(t_b f_j), out[b, half + j] = cos(t_b f_j), half = dim/2;
 * dim odd -> last column 0 (models/layers.py:515-529).  The frequency table
 * f_j = exp(-j ln(10000)/(half-1)) is supplied by the host (freqs[half]): t_b reaches 999, so a 1-ulp
 * difference between two exp implementations would be amplified ~1000x in the sin/cos argument. */
int stk_timestep_embedding_f32(const float* t, const float* freqs, float* out, int B, int dim, void* stream);
/* Gaussian Fourier features: p = x_b * W_j * 2 * pi; out[b,j] = sin p, out[b,nf+j] = cos p */
int stk_fourier_embedding_f32(const float* x, const float* W, float* out, int B, int nf, void* stream);
/* out[n,:] = a[n]*x[n,:] + s[n]*z[n,:]   (perturbation kernel x_t = mean + std*z, losses.py:118-119) */
int stk_perturb_f32(const float* x, const float* z, const float* a, const float* s, float* out,
                    int N, long inner, void* stream);
/* Per-sample score-matching loss (losses.py:122-132) from the raw network output `net`:
 *   score = vp ? -net/std[n] : net ;  r = score*std[n] + z           (mode 0)
 *                                      r = score + z/std[n]           (mode 1, likelihood weighting)
 *   loss[n] = wgt[n] * red * sum(r^2),  red = reduce_mean ? 1/inner : 0.5
 * bwd: dnet = dloss[n] * d loss[n] / d net  (written). */
int stk_sm_loss_fwd_f32(const float* net, const float* z, const float* std, const float* wgt,
                        float* loss, int N, long inner, int vp, int mode, int reduce_mean, void* stream);
int stk_sm_loss_bwd_f32(const float* net, const float* z, const float* std, const float* wgt,
                        const float* dloss, float* dnet, int N, long inner, int vp, int mode,
                        int reduce_mean, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer side (flat fp32 buffers of n elements).
 *   stk_sumsq_f32:  out[0] = sum x^2 (deterministic two-stage; ws >= 1024 floats)
 *   stk_adam_f32:   torch.optim.Adam/AdamW single-tensor update (losses.py:29-41,47-56) with the
 *                   clip_grad_norm_ coefficient min(1, max_norm/(sqrt(*sumsq)+1e-6)) applied to g on
 *                   the fly when sumsq != NULL and max_norm >= 0; g itself is rescaled in place
 *                   (as clip_grad_norm_ does).  bc1 = 1-b1^t, bc2 = 1-b2^t.
 *   stk_ema_f32:    s -= one_minus_decay * (s - p)      (models/ema.py:50-51)
 * ------------------------------------------------------------------------------------------ */
int stk_sumsq_f32(const float* x, long n, float* out, float* ws, void* stream);
int stk_adam_f32(float* p, float* g, float* m, float* v, long n,
                 float lr, float b1, float b2, float eps, float weight_decay, int adamw,
                 float bc1, float bc2, const float* sumsq, float max_norm, void* stream);
/* stk_adam_f32 with AMSGrad (torch.optim.Adam(amsgrad=True), losses.py:33): vmax = max(vmax, v), the denominator uses vmax */
int stk_adam_amsgrad_f32(float* p, float* g, float* m, float* v, float* vmax, long n,
                         float lr, float b1, float b2, float eps, float weight_decay, int adamw,
                         float bc1, float bc2, const float* sumsq, float max_norm, void* stream);
int stk_ema_f32(float* shadow, const float* p, long n, float one_minus_decay, void* stream);

/* Sample post-processing, replaces `np.clip(samples.permute(0,2,3,1).cpu().numpy() * 255., 0, 255).astype(np.uint8)`
 * (sampling_lib.py:43): out[n, hw, c] = uint8(clip(255 * x[n, c, hw], 0, 255)), float NCHW -> uint8 NHWC on the device,
 * so that 1/4 of the bytes cross PCIe. */
int stk_samples_to_uint8(const float* x, unsigned char* out, int N, int C, long HW, void* stream);

/* Input-pipeline tail on the device (datasets.py:313-324, run_lib.py:72-75): uint8 NHWC images -> float NCHW batch.
 *   v = u8 * (1/255);  flip != 0: image n is mirrored left-right iff stk_uniform(seed ^ 0x5DEECE66D, n) < 0.5;
 *   dequant != 0: v = (255 v + stk_uniform(seed, output element index)) / 256;  centered != 0: v = 2 v - 1. */
int stk_preprocess_u8(const unsigned char* img, float* out, int N, int C, int H, int W, int flip, int dequant,
                      int centered, unsigned long long seed, void* stream);

/* Counter-based RNG shared by both libraries so dropout masks are reproducible across them:
 * u(seed, i) in [0,1) from a 64-bit mix of (seed, i); keep iff u >= p.
 * stk_dropout_mask_f32 materialises mask[i] = keep ? 1/(1-p) : 0 (debug / oracle use). */
int stk_dropout_mask_f32(float* mask, long n, float p, unsigned long long seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STK_H */
