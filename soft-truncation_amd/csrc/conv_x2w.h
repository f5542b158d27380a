// conv_x2w.h -- 3x3 / stride 1 / pad 1 weight gradient on the fp16 two-way split with BOTH operands read as planes
// (conv_pl.h), staged by LDS-DMA and fed to the MFMAs through the LDS transpose read, for gfx950.
// Included by conv.hip inside its anonymous namespace, after conv_x2d.h.
//
//   dW[co, ci, kh, kw] = sum over pixels p of  dy[co, p] * x[ci, p + (kh - 1) W + (kw - 1)]        (zero outside the map)
//
// i.e. nine GEMMs D = A B with M = co, N = ci and K = PIXELS.  Planes are [pixel][32 channels]: the contraction index is
// the slow one for both operands, exactly the case `ds_read_b64_tr_b16` exists for -- a 16-lane group reads a
// [4 pixels][16 channels] block and every lane receives 4 consecutive PIXELS of its channel (semantics pinned by
// tools/_probe/tr.hip on the hardware).  What that buys over x2::wgrad3_kernel (fp32 NCHW operands, split on the fly,
// one kernel row per workgroup, three shifted windows of x written to LDS per chunk):
//   * a tap shift is a shift by whole 64-byte LDS rows, so ONE staged tile per operand serves all nine taps;
//   * no conversion and no ds_write at all: both tiles arrive by `buffer_load_dwordx4 ... lds`, halo pixels outside the
//     map as zeros from the buffer range check; one barrier per chunk;
//   * the planes are the ones the forward (x) and the data gradient (dy) already use -- no fp32 copy of either tensor
//     is read, so GroupNorm need not write one for its 3x3 consumers.
// Workgroup = 128 output channels (wave w owns 32-block w) x 32 input channels x 9 taps, over a slab of the pixels;
// partial slabs [tap][Cout][Cin] are summed by splitk_reduce_kernel in a fixed order, as for the other weight gradients.
//
// Round 6 -- where the taps' shifts sit.  Rounds 2-5 shifted x for all nine taps: per 16-pixel half chunk a wave read 4 fragments
// of dy and 36 of x (9 taps x 2 planes x 2 reads) for 27 MFMAs.  But
//   sum_p dy[co, (py, px)] x[ci, (py + kh - 1, px + kw - 1)]  =  sum_q dy[co, (qy, qx - (kw - 1))] x[ci, (qy + kh - 1, qx)]
// (q = p moved by the tap's column offset; dy = 0 where qx - (kw - 1) leaves the map -- exactly the pairs the left side drops
// because x is outside): the COLUMN shift can sit on dy and only the ROW shift on x.  Three column-shifted fragments of dy and
// three row-shifted fragments of x serve the nine taps: 12 + 12 fragment reads per half chunk instead of 4 + 36 (-40 % of the
// LDS read traffic, 209 instead of 235 registers), the same 27 MFMAs.  The dy tile of a 32-channel block gets one halo column
// on each side ([ROWS][COLS + 2] pixel slots, <= 48: three DMA instructions per plane instead of two), the x tile loses its
// halo columns ([ROWS + 2][COLS] slots, <= 96: six instructions per plane instead of seven).  Measured (MI355X, batch 128,
// kernel + reduce, 256 workgroups, profiles/r06_x2w.txt): 128 -> 128 @ 32 x 32 138.8 -> 132.9 us, 384 -> 128 397.3 -> 373.5, the
// 16- / 8- / 4-wide maps within 2 %; in the step -0.04 ms: with ONE workgroup per CU the kernel is bound by the latency of its
// LDS-DMA -- or so it seemed: a third LDS stage bought nothing either (STAGES below).
#pragma once

namespace x2w {

typedef short s4 __attribute__((__vector_size__(8)));
typedef __attribute__((address_space(3))) s4 lds_s4;
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
  const unsigned char* dypl; const float* dyrec; long dy_ps;     // planes of dy [N, Cout, H, W], scale record, bytes per plane
  const unsigned char* xpl; const float* xrec; long x_ps;        // planes of x  [N, Cin,  H, W]
  float* part; long part_stride;                                 // slabs [split][tap][Cout][Cin]
  int N, H, W, HW, Cin, Cout, Cob, Cib;                          // Cob / Cib: 32-channel blocks
  int tiles_co, tiles_ci, nchunks_total, chunks_per_split;
};

__device__ __forceinline__ halfx8 cat(s4 lo, s4 hi) {
  typedef short s8 __attribute__((__vector_size__(16)));
  const s8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(halfx8, v);
}

constexpr int A_INSTR = 3;
constexpr int A_BLK = A_INSTR * 16 * 64;    // 3072: 48 pixel slots of one 32-channel block of dy
constexpr int A_PLANE = 4 * A_BLK;          // 12288
constexpr int B_INSTR = 6;
constexpr int B_PLANE = B_INSTR * 16 * 64;  // 6144: 96 pixel slots of x
constexpr int BUF = 2 * A_PLANE + 2 * B_PLANE;   // 36864 per stage

// COLS = min(W, 32): pixels of a chunk per map row (a chunk is 32 consecutive pixels = 32 / COLS whole rows, or a 32-pixel
// piece of one row when W > 32; COLS = 4: two whole 4 x 4 images).
// GROUPS = 2 (round 4): a workgroup of EIGHT waves, two groups of four that own the SAME 128 x 32 x 9 block and take the
// split's chunks alternately, each with its own LDS stages; after the last chunk group 1 hands its 144 accumulators per thread
// to group 0 through the (now free) LDS in three rounds of three taps, and group 0 writes the slab: half the K split and slab
// traffic at the same eight waves per CU.  The sum of the two groups is taken in a fixed order (group 0 + group 1).
// STAGES = 2: LDS stages of the DMA ring.  Three stages (two chunks in flight, 110 KB of LDS) were measured in round 6 and bought
// nothing -- 128 -> 128 @ 32 x 32 at 256 workgroups 134.4 (two) / 135.9 us (three), 384 -> 128 378.2 / 380.8 -- and cost the step
// 0.2 ms (35.82 -> 35.99 / 36.04 ms: more LDS, 296 instead of 209 registers, less room for the main chain's waves on the CU), so
// the kernel is not waiting for its DMA either (profiles/r06_x2w.txt).  Also measured: the next chunk's nine DMA instructions issued
// BETWEEN the MFMA groups of the current chunk instead of in a burst in front of them (an LDS-DMA costs its wave 60-180 cycles of issue,
// and a SIMD holds ONE wave of this kernel in the two-stream geometry): seven times SLOWER, because hipcc then puts an s_waitcnt vmcnt(0)
// in front of every DMA and every following fragment read (it cannot prove that the DMA's LDS bytes are not the ones being read, even
// with the two stages as two __shared__ objects), which makes every DMA synchronous.  The burst at the chunk boundary stays.
template <int COLS, int GROUPS = 1>
__global__ __launch_bounds__(256 * GROUPS, 2) void wgrad_kernel(Args a) {
  constexpr int STAGES = 2;
  constexpr bool TWO = COLS == 4;           // 4 x 4 maps: a chunk is two whole images
  constexpr int ROWS = 32 / COLS;           // chunk rows (TWO: 8 = 4 + 4)
  constexpr int TPA = COLS + 2;             // dy tile: row pitch in slots (halo column left and right)
  constexpr int TRB = TWO ? 12 : ROWS + 2;  // x tile: rows (halo row above and below; TWO: 6 per image)
  constexpr int LDS = STAGES * BUF;
  static_assert(ROWS * TPA <= A_INSTR * 16 && TRB * COLS <= B_INSTR * 16, "tiles exceed the staged slots");
  __shared__ __attribute__((aligned(1024))) unsigned char lds_all[LDS * GROUPS];
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int grp = GROUPS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const lds = lds_all + grp * LDS;               // this group's stages
  // the operands' scales are needed by the slab store only: their partial maxima are LOADED here and reduced behind the main loop
  // (every group for itself, in LDS of its own: the barrier inside is workgroup-wide)
  __shared__ float red_amax[2][GROUPS][4];
  x2::LateAmax amax_dy, amax_x;
  amax_dy.load(a.dyrec, x2::NPART, tid);
  amax_x.load(a.xrec, x2::NPART, tid);
  const int ntiles = a.tiles_co * a.tiles_ci;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id % ntiles, zs = id / ntiles;
  const int tco = tile % a.tiles_co, tci = tile / a.tiles_co;
  const int co_blk = tco * 4 + wid;                                     // this wave's 32 output channels
  const bool co_live = co_blk < a.Cob;
  const int c_begin = zs * a.chunks_per_split;
  const int c_last = min(a.nchunks_total, c_begin + a.chunks_per_split) - 1;

  // ---- DMA sources ----------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t a_rs = x3::make_rsrc(a.dypl, 2L * a.dy_ps);
  const __amdgpu_buffer_rsrc_t b_rs = x3::make_rsrc(a.xpl, 2L * a.x_ps);
  const int piece = (lane & 3) * 16;
  // dy tile of this wave's block: slot s = 16 j + lane / 4 = (row r, column c) holds map pixel (y0 + r, x0 + c - 1)
  int a_rel[A_INSTR], a_c[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int sl = 16 * j + (lane >> 2);
    int r = sl / TPA;
    a_c[j] = sl - r * TPA;
    int img = 0;
    if (TWO) { img = r >> 2; r &= 3; }                                  // next image: Cob blocks on
    if (sl >= ROWS * TPA || !co_live) a_c[j] = -1000000;                // never inside the map
    a_rel[j] = (img * a.Cob * a.HW + r * a.W + a_c[j] - 1) * 64 + piece;
  }
  // x tile: slot s = 16 j + lane / 4 = (row ty, column tx) holds map pixel (y0 - 1 + ty, x0 + tx).  The 12 instructions (6 per
  // plane) go round the four waves, three each -- every wave issues the same number per stage (the vmcnt of the ring)
  int b_rel[3], b_ty[3], b_dst[3]; unsigned b_pl[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int idx = wid + 4 * u, j = idx % B_INSTR, pl = idx / B_INSTR;
    const int sl = 16 * j + (lane >> 2);
    b_ty[u] = sl / COLS;
    const int tx = sl - b_ty[u] * COLS;
    int img = 0;
    if (TWO) { img = b_ty[u] / 6; b_ty[u] -= 6 * img; }                 // row inside its image's tile
    if (sl >= TRB * COLS) b_ty[u] = -1000000;
    b_rel[u] = (img * a.Cib * a.HW + (b_ty[u] - 1) * a.W + tx) * 64 + piece;
    b_dst[u] = 2 * A_PLANE + pl * B_PLANE + j * 1024;
    b_pl[u] = (unsigned)pl;
  }
  auto stage = [&](int c, unsigned char* buf) {
    const int p0 = c * 32;                                               // first pixel of the chunk, over N * HW
    const int b = p0 / a.HW, hw0 = p0 - b * a.HW;
    const int y0 = hw0 / a.W, x0 = hw0 - y0 * a.W;
    const int a_chunk = ((b * a.Cob + co_blk) * a.HW + hw0) * 64;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const int xx = x0 + a_c[j] - 1;
      const unsigned vo = (xx >= 0 && xx < a.W) ? (unsigned)(a_chunk + a_rel[j]) : 0x80000000u;    // outside the map: DMA of zeros
#pragma unroll
      for (int s = 0; s < 2; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_void*)(buf + s * A_PLANE + wid * A_BLK + j * 1024), 16,
                                                 (int)vo, (int)(s * (unsigned)a.dy_ps), 0, 0);
    }
    const int b_chunk = ((b * a.Cib + tci) * a.HW + hw0) * 64;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int yy = y0 + b_ty[u] - 1;
      const unsigned vo = (yy >= 0 && yy < a.H) ? (unsigned)(b_chunk + b_rel[u]) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_void*)(buf + b_dst[u]), 16, (int)vo, (int)(b_pl[u] * (unsigned)a.x_ps), 0, 0);
    }
  };

  floatx16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // ---- transpose-read addresses: lane (m = lane % 16, g = lane / 16) supplies the 8 bytes at pixel k0 + m / 4, channels
  // r0 + 4 (m % 4) .. + 3 and receives pixels k0 .. k0 + 3 of channel r0 + m:  r0 = 16 (g & 1), k0 = 8 (g >> 1) + 4 h + 16 kk
  const int m = lane & 15, g = lane >> 4;
  const int ch_off = (16 * (g & 1) + 4 * (m & 3)) * 2;
  int a_off[2][2], b_off[2][2];                                          // [kk][h]: tap (1, 1) of the pixel this lane addresses
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 16 * kk + 8 * (g >> 1) + 4 * h + (m >> 2);           // pixel of the chunk
      const int krow = k / COLS, kcol = k - krow * COLS;
      a_off[kk][h] = wid * A_BLK + (krow * TPA + kcol + 1) * 64 + ch_off;
      const int trow = TWO ? (krow >> 2) * 6 + (krow & 3) : krow;        // two images: rows 0..3 | 4..7 -> tiles 0 | 1
      b_off[kk][h] = 2 * A_PLANE + ((trow + 1) * COLS + kcol) * 64 + ch_off;
    }
  constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};                    // cross terms first (fixed accumulation order)

  auto compute = [&](const unsigned char* buf) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      halfx8 af[3][2];                                                   // [kw][plane]: dy at column qx - (kw - 1)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          af[t][s] = cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(buf + s * A_PLANE + a_off[kk][0] - (t - 1) * 64)),
                         __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(buf + s * A_PLANE + a_off[kk][1] - (t - 1) * 64)));
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) {                                   // one kernel row: x at row qy + (kh - 1)
        halfx8 bf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
          bf[s] = cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(buf + s * B_PLANE + b_off[kk][0] + (t3 - 1) * COLS * 64)),
                      __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(buf + s * B_PLANE + b_off[kk][1] + (t3 - 1) * COLS * 64)));
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int t = 0; t < 3; ++t)
            acc[3 * t3 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t][SA[pr]], bf[SB[pr]], acc[3 * t3 + t], 0, 0, 0);
      }
    }
  };

  // group g takes chunks c_begin + g, c_begin + g + GROUPS, ...; both groups run the same number of barrier rounds.  The ring keeps
  // STAGES - 1 chunks in flight: chunk i of this group lives in stage i % STAGES.
  const int rounds = (c_last - c_begin + GROUPS) / GROUPS;
#pragma unroll
  for (int i = 0; i < STAGES - 1; ++i)
    if (c_begin + i * GROUPS + grp <= c_last) stage(c_begin + i * GROUPS + grp, lds + i * BUF);
  int cur = 0;
  for (int it = 0; it < rounds; ++it) {
    const int c = c_begin + it * GROUPS + grp;
    // this wave's part of chunk c has landed ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // ... everybody's has, and nobody reads the stage refilled below any more
    const int nxt = cur + STAGES - 1 >= STAGES ? cur - 1 : cur + STAGES - 1;
    if (c + (STAGES - 1) * GROUPS <= c_last) stage(c + (STAGES - 1) * GROUPS, lds + nxt * BUF);
    if (c <= c_last) compute(lds + cur * BUF);
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  }
  const float sa = x2::pow2_scale_of(amax_dy.reduce(red_amax[0][grp], tid));
  const float sb = x2::pow2_scale_of(amax_x.reduce(red_amax[1][grp], tid));
  const float unscale = 1.f / (sa * sb);
  if (GROUPS == 2) {
    // group 1 -> group 0, three taps (48 floats per thread, 48 KB) per round through group 0's stages; element (t, e) of
    // thread tid at float (16 t + e) 256 + tid: conflict-free for writer and reader
    float* const xch = reinterpret_cast<float*>(lds_all);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      __syncthreads();                                 // the tiles (r = 0) / the previous round's values have been consumed
      if (grp == 1) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) xch[(16 * t + e) * 256 + tid] = acc[3 * r + t][e];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[3 * r + t][e] += xch[(16 * t + e) * 256 + tid];
      }
    }
    if (grp == 1) return;
  }

  // partial slab of split zs as [tap][Cout][Cin] (lanes = ci, contiguous); splitk_reduce_kernel re-lays it out
  float* slab = a.part + (long)zs * a.part_stride;
  const int fk = lane >> 5, fc = lane & 31;
  const int ci = tci * 32 + fc;
  if (co_live && ci < a.Cin) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co_blk * 32 + 4 * fk + igemm::strip_row(e);
        if (co < a.Cout) slab[((long)t * a.Cout + co) * a.Cin + ci] = unscale * acc[t][e];
      }
  }
}

// What bounds this kernel ALONE is its LDS-DMA traffic: chunks x tiles x 36 KB per launch pass from the Infinity Cache into LDS (both
// operands are activations; every 128 x 32 tile re-reads all of dy), and every shape lands at 4.5-5.2 TB/s of it.  A 128 x 64-tile variant
// (eight waves sharing one dy tile, -33 % bytes per MFMA) confirmed it -- 133 -> 121 us, 376 -> 317, 0.455 of 833 on 512 -> 256 @ 16 x 16 --
// and LOST inside the step at every workgroup count (+0.5 ... +3 ms): eight 214-register waves and 98 KB of LDS leave no room on a CU for the
// main chain's waves.  Measured and removed in round 6 (profiles/r06_x2w.txt block 4; git history).  (A register diet to 192 per lane, so that
// TWO of the main chain's 160-register GEMM waves fit beside one of this kernel's on a SIMD, did not get past the compiler: it fills the
// 256-register budget of the launch bounds with hoisted fragment reads whatever the source says, and ignores amdgpu_num_vgpr.)

// H = W a power of two >= 8 (a chunk of 32 pixels is whole rows or a piece of one row, never across images) or H = W = 4
// (a chunk is two whole images), channel counts in whole 32-blocks (planes), enough work to fill the chip
struct Plan { int ok; int splits; int chunks_per_split; long slab; int groups; };
inline int groups_mode() { static const int v = [] { const char* e = getenv("STK_X2W_GROUPS"); return e ? atoi(e) : 2; }(); return v; }
// wgs: workgroups the K split fills; 0 = the default for a launch that shares the chip with another stream (STK_X2W_WGS, 256), else the
// caller's figure (the engine passes STK_X2W_WGS_ALONE = 512 when the weight gradient runs on the main stream with nothing beside it)
inline Plan plan(int N, int H, int W, int Cin, int Cout, int wgs = 0) {
  Plan r = {0, 1, 0, 0, 1};
  if (H != W || W < 4 || (W & (W - 1)) || Cin % 32 || Cout % 32 || Cin < 32 || Cout < 32) return r;
  const long px = (long)N * H * W;
  if (px % 32 || px / 32 > 0x7fffffffL / 64) return r;
  const int nch = (int)(px / 32);
  const long tiles = (long)stk_cdiv(Cout, 128) * (Cin / 32);
  // two workgroups per CU; >= 8 chunks per workgroup; and a cap on the slab traffic (every split writes, and the
  // reduce reads, 9 Cout Cin floats): STK_WGRAD_SLAB_MB, default 128 (A/B on one box: 16 MB 55.7 ms per step, 32 MB 48.0, 64 MB 46.3, 128 MB 46.1 -- the parallelism of the K split is worth more than its slab traffic)
  static const long cap_mb = [] { const char* e = getenv("STK_WGRAD_SLAB_MB"); return e ? atol(e) : 128L; }();
  // at most 512 workgroups = ONE round of two per CU (rounding the quotient up gave 516 for 12 tiles: a second round of four
  // workgroups, 441 us instead of ~330 on the 384 -> 128 layer at 32 x 32)
  // STK_X2W_WGS (A/B): workgroups the K split fills.  Round 5: 256 = ONE four-wave workgroup per CU (128 eight-wave ones on the 8- / 16-wide
  // maps), not the two that fit.  The weight gradients run on the side stream BESIDE the main chain of the backward; two workgroups per CU
  // with their 256 accumulation registers per lane leave no room for anybody else's waves, so "two streams" was time slicing (summed
  // kernel durations 54 ms for 40 ms of work, tools/stream_timeline.py); with half of every CU free the main chain's kernels really
  // run beside them -- and the slab traffic halves.  Measured inside the step (512 -> 256, shortcut convolutions' backward on the main stream; profiles/
  // r05_insitu_sweeps.txt): CIFAR-10 net 37.6 -> 36.4 ms, 256x256 net at batch 4 39.4 -> 37.6, 64x64 net 121.6 -> 120.4.  The kernel
  // alone is slower that way (the round-3 micro-benchmark chose 512); the step is what counts.
  static const long shared = [] { const char* e = getenv("STK_X2W_WGS"); return e && atol(e) > 0 ? atol(e) : 256L; }();
  const long target = wgs > 0 ? wgs : shared;
  long splits = tiles >= target ? 1 : target / tiles;
  if (splits > nch / 8) splits = nch / 8;
  const long cap = (cap_mb << 20) / (9L * Cout * Cin * 4);
  if (splits > cap) splits = cap;
  if (splits < 1) splits = 1;
  // two groups of four waves per workgroup (wgrad_kernel<COLS, 2>): half the splits, the same eight waves per CU, when every
  // group still gets >= 8 chunks (STK_X2W_GROUPS=1: off, 3: every map width).  Measured, kernel + reduce, batch 128, us, one
  // group -> two (profiles/r04_x2w_groups.txt): 16 x 16: 115.9 -> 113.1, 203.8 -> 199.3, 73.7 -> 68.0; 8 x 8: 52.1 -> 47.8, 71.7 ->
  // 68.6; but 32 x 32: 122.8 -> 126.4, 317.9 -> 339.0 (eight waves in barrier lockstep) and 4 x 4: 32.8 -> 35.5: the 8- and
  // 16-wide maps only.
  const bool gw = groups_mode() == 3 || (groups_mode() == 2 && (W == 8 || W == 16));
  if (gw && splits >= 2 && splits % 2 == 0 && nch / splits >= 8) { r.groups = 2; splits /= 2; }
  r.chunks_per_split = stk_cdiv(nch, splits);
  r.splits = stk_cdiv(nch, r.chunks_per_split);
  r.slab = 9L * Cout * Cin;
  r.ok = nch >= 8;
  return r;
}

}  // namespace x2w
