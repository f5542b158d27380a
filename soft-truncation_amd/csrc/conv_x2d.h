// conv_x2d.h -- the fp16 two-way-split convolution (conv_x2.h) with BOTH operands staged by LDS-DMA
// (`buffer_load_dwordx4 ... lds`), for gfx950.  Included by conv.hip inside its anonymous namespace, after conv_pl.h.
//
// Why.  With plane operands (conv_pl.h) the staging of x2::gemm_kernel is pure copying, yet the kernel stayed at 0.31
// of the 3-MFMA ceiling.  Ablations on the 128 -> 128 3x3 layer at 32x32, batch 128 (profiles/r02_conv_ablation.txt):
//     full kernel 148.9 us | without the ds_write_b128 of the staging 120.6 | MFMAs + barriers only 115.7.
// A ds_write_b128 moves 5 source VGPRs to the LDS at 2 cycles per dword and wave (MI355X_MICROARCH.md, LDS): the 8
// stores per thread and chunk keep the LDS busy for ~830 of the 1536 cycles the two resident workgroups' MFMAs of a
// chunk take, on top of ~510 cycles of operand reads.  LDS-DMA writes the LDS from the memory pipe instead -- no
// staging registers, no store instructions -- and the freed registers buy a 128 x 256 tile (wave tile 64 x 128:
// one weight tile now feeds twice the MFMAs, 0.5 instead of 0.67 operand reads per MFMA, and the 32x32 layers of
// batch 128 become ONE round of 512 workgroups instead of two with their prologue / epilogue bursts).
//
// LDS image of an operand tile: [split][row][64 bytes = 32 k], NO padding (a DMA instruction writes wave-uniform base
// + 16 lane: 16 rows x 64 bytes), bank conflicts removed by an XOR swizzle of the 16-byte slots inside a row,
//     slot(row, seg) = seg ^ ((row >> 2) & 3),
// applied on the SOURCE address of the DMA and on the address of the operand read (cdna_hip_programming.md, rule 21):
// any 32 consecutive rows read with one `seg` then cover every bank group exactly once per ds_read_b128 lane group.
#pragma once

namespace x2d {

using x3::KC;
typedef __attribute__((address_space(3))) void lds_void;

template <int TN>
struct Geo {
  static constexpr int A_PLANE = 128 * 64, A_BYTES = 2 * A_PLANE;
  static constexpr int B_PLANE = TN * 64, B_BYTES = 2 * B_PLANE;
  static constexpr int LDS = A_BYTES + B_BYTES;                 // 32 KB (TN = 128) / 48 KB (TN = 256)
  static constexpr int NJ = TN / 64;                            // 32-pixel MFMA column blocks per wave (2 waves along N)
  static constexpr int BBLK = TN / 64;                          // 16-row DMA blocks of B per wave and plane
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* dst, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, (int)voff, (int)soff, 0, 0);
}

// (Tried and dropped: a static wave priority per workgroup generation, s_setprio by (blockIdx >> 8) & 1, to break
// the lockstep of the two workgroups of a CU -- 132 -> 142 us on the 128 -> 128 layer.  Reading ALL fragments of a chunk
// before its first MFMA, so that the tile is free and the next chunk's DMA in flight for 48 instead of 24 MFMAs: 250
// VGPRs, the training step 43.9 -> 44.5 ms.)
//
// NBUF = 2 (round 3): two LDS images of the tile pair.  The DMA of chunk c + 1 is issued right after the barrier that
// opens chunk c, into the buffer chunk c - 1 was read from, so it has ALL 24 MFMAs of the chunk (and the partner
// workgroup's) to land instead of the second half's 12, and the mid-chunk barrier goes: one barrier per chunk.
// 2 x 32 KB x 2 workgroups per CU = 128 KB of the 160 KB.
// ABL (ablation, benchmarks only -- results are garbage unless 0): bit 0 stages B only for the first tap of a channel
// group, bit 1 stages A only for the first chunk, bit 2 reads the operand fragments only in the first chunk.
template <int TAPS, int TN, class EP, int NBUF = 1, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(ConvP p, x3::Src q, int M, int Nn, int tiles_m, int tiles_n,
                                                      int nchunks_total, int chunks_per_split,
                                                      const float* __restrict__ xpart, int nxpart) {
  using G = Geo<TN>;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[G::LDS * NBUF];
  unsigned char* const As = lds;
  unsigned char* const Bs = lds + G::A_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  __shared__ float red_amax[4];
  x2::LateAmax xmax;                                    // loads now, reduction behind the main loop (the epilogue's unscale needs it)
  xmax.load(xpart, nxpart, tid);
  const float sw = x2::weight_scale(q.wp);
  const int ntiles = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id % ntiles, zs = id / ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * TN;
  const int c_begin = zs * chunks_per_split;
  const int c_last = min(nchunks_total, c_begin + chunks_per_split) - 1;

  // ---- DMA sources.  A lane of a DMA instruction owns (row = 16-row block base + lane / 4, slot = lane % 4) and
  // fetches segment slot ^ f(row); every block base is a multiple of 16, so f(row) = (lane >> 4) & 3 for all of them.
  const unsigned seg_src = (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
  // A: prepared weights, [split][k / 32][tap][row (Mpad)][32] fp16 behind the header; wave w stages rows 32 w .. 32 w + 31
  const unsigned a_plane2 = (unsigned)q.taps * q.Mpad * q.Kc * 2u;
  const unsigned a_chunk2 = (unsigned)q.Mpad * KC * 2u;
  const __amdgpu_buffer_rsrc_t a_rs = x3::make_rsrc(reinterpret_cast<const unsigned char*>(q.wp) + x2::HEADER, 2L * a_plane2);
  const unsigned a_voff = (unsigned)(m0 + 32 * wid + (lane >> 2)) * 64u + seg_src;
  // B: planes [split][n][cb][pixel][32]; wave w stages rows (TN / 4) w .. + TN / 4 - 1 in blocks of 16
  const __amdgpu_buffer_rsrc_t b_rs = x3::make_rsrc(q.pl, 2L * q.pl_stride);
  const unsigned b_ps = (unsigned)q.pl_stride;
  unsigned b_base[G::BBLK], b_mask[G::BBLK];
  {
    const int Cb = q.Kc >> 5;
#pragma unroll
    for (int jb = 0; jb < G::BBLK; ++jb) {
      const int n = n0 + (TN / 4) * wid + 16 * jb + (lane >> 2);
      b_mask[jb] = 0; b_base[jb] = 0;
      if (n < Nn) {
        const int b = n / p.HW, hw = n - b * p.HW;
        const int y = hw / p.W, x = hw - y * p.W;
        if (TAPS == 1) b_mask[jb] = 1u;
#pragma unroll
        for (int t = 0; t < (TAPS == 9 ? 9 : 0); ++t) {
          const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
          if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) b_mask[jb] |= 1u << t;
        }
        b_base[jb] = ((unsigned)(b * Cb) * p.HW + hw) * 64u + seg_src;
      }
    }
  }
  auto stage = [&](int c, int buf = 0) {
    const int cc = TAPS == 9 ? c / 9 : c, tap = c - cc * TAPS;           // scalar
    const unsigned a_soff = (unsigned)c * a_chunk2;
    unsigned char* const As = lds + buf * G::LDS;
    unsigned char* const Bs = As + G::A_BYTES;
    if (!(ABL & 2) || c == c_begin) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        dma16(a_rs, As + s * G::A_PLANE + (32 * wid + 16 * h) * 64, a_voff + h * 1024u, a_soff + s * a_plane2);
    }
    const int shift = TAPS == 9 ? ((tap / 3 - 1) * p.W + (tap % 3 - 1)) * 64 : 0;
    const unsigned b_soff = (unsigned)cc * (unsigned)p.HW * 64u;
    if ((ABL & 1) && tap != 0) return;
#pragma unroll
    for (int jb = 0; jb < G::BBLK; ++jb) {
      const unsigned dead = (((b_mask[jb] >> tap) & 1u) ^ 1u) << 31;      // halo / out-of-range rows: DMA of zeros
      const unsigned vo = (b_base[jb] + (unsigned)shift) | dead;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        dma16(b_rs, Bs + s * G::B_PLANE + ((TN / 4) * wid + 16 * jb) * 64, vo, b_soff + s * b_ps);
    }
  };

  EP ep;
  ep.preload(p, m0, n0, TN, M, Nn, tid);

  floatx16 acc[2][G::NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < G::NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm0 = (wid & 1) * 64, wn0 = (wid >> 1) * (TN / 2);
  const int fk = lane >> 5, fc = lane & 31;
  const int fsw = (fc >> 2) & 3;                                          // f(row) of the operand rows this lane reads
  const unsigned char* a_rd = As + (wm0 + fc) * 64;
  const unsigned char* b_rd = Bs + (wn0 + fc) * 64;
  const int ko0 = ((0 + fk) ^ fsw) * 16, ko1 = ((2 + fk) ^ fsw) * 16;     // slot of k segment (2 kk + fk)

#define STK_D_FRAGS_AT(KO, OFF)                                                                           \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
      a[i][s] = *reinterpret_cast<const halfx8*>(a_rd + (OFF) + s * G::A_PLANE + i * 32 * 64 + (KO));      \
    _Pragma("unroll") for (int j = 0; j < G::NJ; ++j)                                                       \
      b[j][s] = *reinterpret_cast<const halfx8*>(b_rd + (OFF) + s * G::B_PLANE + j * 32 * 64 + (KO));      \
  }
#define STK_D_FRAGS(KO) STK_D_FRAGS_AT(KO, 0)
  // three products per tile, the two cross terms first (fixed accumulation order)
  constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};
#define STK_D_MFMAS                                                                                        \
  _Pragma("unroll") for (int pr = 0; pr < 3; ++pr) _Pragma("unroll") for (int i = 0; i < 2; ++i)             \
    _Pragma("unroll") for (int j = 0; j < G::NJ; ++j) {                                                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][SA[pr]], b[j][SB[pr]], acc[i][j], 0, 0, 0);   \
    }

  halfx8 a[2][2], b[G::NJ][2];
  stage(c_begin);
  if constexpr (NBUF == 2) {
    int cur = 0;
    for (int c = c_begin; c <= c_last; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of chunk c has landed ...
      __syncthreads();                                   // ... so has everybody else's, and nobody reads chunk c - 1 any more
      if (c < c_last) stage(c + 1, cur ^ 1);             // chunk c + 1 streams into the other buffer under this chunk's MFMAs
      const int off = cur * G::LDS;
      STK_D_FRAGS_AT(ko0, off)
      STK_D_MFMAS
      STK_D_FRAGS_AT(ko1, off)
      STK_D_MFMAS
      cur ^= 1;
    }
  } else {
    for (int c = c_begin; c <= c_last; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of chunk c has landed ...
      __syncthreads();                                   // ... and so has everybody else's
      if (!(ABL & 4) || c == c_begin) { STK_D_FRAGS(ko0) }
      STK_D_MFMAS
      if (!(ABL & 4) || c == c_begin) { STK_D_FRAGS(ko1) }
      __syncthreads();                                   // nobody reads the tile any more (the barrier waits lgkmcnt(0))
      if (c < c_last) stage(c + 1);                      // chunk c + 1 streams in under the second half's MFMAs
      STK_D_MFMAS
    }
  }
#undef STK_D_MFMAS
#undef STK_D_FRAGS
#undef STK_D_FRAGS_AT

  const float unscale = 1.f / (sw * x2::pow2_scale_of(xmax.reduce(red_amax, tid)));
  ep.stage(lds, tid);
  ep.init(p, 0, zs);
#pragma unroll
  for (int j = 0; j < G::NJ; ++j) {
    const int n = n0 + wn0 + j * 32 + fc;
    const bool nok = n < Nn;
    ep.col(p, nok ? n : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] *= unscale;
      ep.strip(p, m0 + wm0 + i * 32 + 4 * fk, M, nok, nok ? n : 0, acc[i][j]);
    }
  }
  ep.finish(p, lds, tid);
}

// ---- 3x3 with ONE staged halo tile of the activations per 32-channel group (round 3) ---------------------------------
// Ablations of gemm_kernel on the data gradient (128 -> 128 at 32 x 32 / 256 -> 256 at 16 x 16 / 384 -> 128 at 32 x 32,
// batch 128; profiles/r03_x2d_ablation.txt): full 133.6 / 115.0 / 320.4 us; B staged for the first tap of a group only
// 122.6 / 102.3 / 296.5; no A staging 125.3 / 108.9 / 300.9; no staging at all 117.3 / 95.9 / 275.7; MFMAs + barriers only
// 109.3 / 91.4 / 257.4.  An LDS-DMA instruction costs the issuing wave 60-180 cycles (MI355X_MICROARCH.md), and the
// nine taps of a channel group re-stage the SAME activations at nine pixel shifts: 8 DMA instructions per wave and chunk
// against 24 MFMAs.  Here a 128-pixel tile (R = 128 / W whole map rows) stages, once per channel group, its halo tile
// [(R + 2) rows][W + 2 columns] x 64 bytes per plane -- zero columns left and right, zero rows outside the map, all
// from the buffer range check -- and the nine taps read their fragments from it at row offsets dy (W + 2) + dx:
// 26 instead of 72 B-side DMA instructions per group and workgroup (W = 32), issued one per wave and chunk into the
// OTHER of two B buffers while the current group computes.  A (weights) is staged per chunk as before.
// Shapes: W = 16, 32 or (NB = 1) 64, maps of whole tiles (H W % 128 == 0), no K split.
// Only NB = 1 (one B buffer, refilled in a burst behind the last tap's fragment reads; three workgroups per CU cover the wait) and
// W = 16 / 32 / 64 are instantiated.  The template still carries the two forms that were measured slower and retired in round 6 --
// NB = 2 (two B buffers, two workgroups per CU) and W = 128 (row strips for the 128- / 256-wide maps: 67 KB of LDS, 187.9 -> 203.1 us)
// -- because they share all of the kernel's code paths but a few constants.
template <int W, class EP, int NB>
__global__ __launch_bounds__(256, 2) void gemm_halo_kernel(ConvP p, x3::Src q, int M, int Nn, int tiles_m, int tiles_n,
                                                           int ngroups, const float* __restrict__ xpart, int nxpart) {
  constexpr int TN = 128, R = TN / W, TP = W + 2, HR = (R + 2) * TP, NI = (HR + 15) / 16, HRP = NI * 16;
  constexpr int A_PLANE = 128 * 64, A_BYTES = 2 * A_PLANE, B_PLANE = HRP * 64, B_BYTES = 2 * B_PLANE;
  constexpr int NK = (NI + 3) / 4;                                        // B instructions per wave and plane
  static_assert(NB == 1 || 2 * NK <= 8, "the B pieces of a group are issued one per chunk over taps 0..7");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[A_BYTES + NB * B_BYTES];
  unsigned char* const As = lds;
  unsigned char* const Bs = lds + A_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  __shared__ float red_amax[4];
  x2::LateAmax xmax;                                    // loads now, reduction behind the main loop
  xmax.load(xpart, nxpart, threadIdx.x);
  const float sw = x2::weight_scale(q.wp);
  const int ntiles = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id % ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * TN;
  const int nchunks = 9 * ngroups;

  const unsigned seg_src = (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
  const unsigned a_plane2 = (unsigned)q.taps * q.Mpad * q.Kc * 2u;
  const unsigned a_chunk2 = (unsigned)q.Mpad * KC * 2u;
  const __amdgpu_buffer_rsrc_t a_rs = x3::make_rsrc(reinterpret_cast<const unsigned char*>(q.wp) + x2::HEADER, 2L * a_plane2);
  const unsigned a_voff = (unsigned)(m0 + 32 * wid + (lane >> 2)) * 64u + seg_src;
  const __amdgpu_buffer_rsrc_t b_rs = x3::make_rsrc(q.pl, 2L * q.pl_stride);
  const unsigned b_ps = (unsigned)q.pl_stride;
  // halo tile rows of this lane: instruction i = wid + 4 k covers LDS rows 16 i .. 16 i + 15, lane -> row 16 i + lane / 4
  unsigned b_voff[NK];
  {
    const int Cb = q.Kc >> 5;
    const int b = n0 / p.HW, hw0 = n0 - b * p.HW, y0 = hw0 / p.W, x0 = hw0 - y0 * p.W;      // x0 = 0 unless the map is wider than a tile
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int row = 16 * (wid + 4 * k) + (lane >> 2);
      const int ty = row / TP, tx = row - ty * TP;
      const int y = y0 - 1 + ty, x = x0 + tx - 1;
      const bool ok = row < HR && y >= 0 && y < p.H && x >= 0 && x < p.W;
      b_voff[k] = ok ? ((unsigned)(b * Cb) * p.HW + (unsigned)(y * p.W + x)) * 64u + seg_src : 0x80000000u;
    }
  }
  auto stage_a = [&](int c) {
    const unsigned a_soff = (unsigned)c * a_chunk2;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        dma16(a_rs, As + s * A_PLANE + (32 * wid + 16 * h) * 64, a_voff + h * 1024u, a_soff + s * a_plane2);
  };
  // piece (k, s) of group cc's halo tile into B buffer `buf`; k is a compile-time constant at every call site
  auto stage_b = [&](int cc, int buf, int k, int s) {
    if (wid + 4 * k >= NI) return;                                        // wave-uniform
    dma16(b_rs, Bs + buf * B_BYTES + s * B_PLANE + (wid + 4 * k) * 1024, b_voff[k],
          (unsigned)cc * (unsigned)p.HW * 64u + s * b_ps);
  };

  EP ep;
  ep.preload(p, m0, n0, TN, M, Nn, tid);

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm0 = (wid & 1) * 64, wn0 = (wid >> 1) * 64;
  const int fk = lane >> 5, fc = lane & 31;
  const int fsw = (fc >> 2) & 3;
  const unsigned char* a_rd = As + (wm0 + fc) * 64;
  const int ka0 = ((0 + fk) ^ fsw) * 16, ka1 = ((2 + fk) ^ fsw) * 16;
  // B: tile pixel wn0 + 32 j + fc = (r, x) -> halo row (r + 1) TP + x + 1 for the centre tap
  int b_row[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pt = wn0 + 32 * j + fc;
    b_row[j] = (pt / W + 1) * TP + (pt % W) + 1;
  }

  halfx8 a[2][2], b[2][2];
  constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};
#define STK_H_MFMAS                                                                                        \
  _Pragma("unroll") for (int pr = 0; pr < 3; ++pr) _Pragma("unroll") for (int i = 0; i < 2; ++i)             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                         \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][SA[pr]], b[j][SB[pr]], acc[i][j], 0, 0, 0);   \
    }
#define STK_H_FRAGS(KA, KSEG)                                                                              \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
      a[i][s] = *reinterpret_cast<const halfx8*>(a_rd + s * A_PLANE + i * 32 * 64 + (KA));                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
      b[j][s] = *reinterpret_cast<const halfx8*>(bbase + s * B_PLANE + brow[j] * 64 + ((((KSEG) + fk) ^ (brow[j] >> 2)) & 3) * 16); \
  }

  // prologue: the whole halo tile of group 0 and the weights of chunk 0
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int s = 0; s < 2; ++s) stage_b(0, 0, k, s);
  stage_a(0);
  int c = 0;
  for (int cc = 0; cc < ngroups; ++cc) {
    const unsigned char* bbase = Bs + (NB == 2 ? (cc & 1) * B_BYTES : 0);
    const bool more = cc + 1 < ngroups;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++c) {
      const int toff = (tap / 3 - 1) * TP + (tap % 3 - 1);
      int brow[2] = {b_row[0] + toff, b_row[1] + toff};
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces (A of chunk c, B pieces issued a chunk ago) landed
      __syncthreads();                                   // ... everybody's
      STK_H_FRAGS(ka0, 0)
      STK_H_MFMAS
      STK_H_FRAGS(ka1, 2)
      __syncthreads();                                   // nobody reads the A tile any more
      if (c + 1 < nchunks) stage_a(c + 1);
      if (NB == 2) {
        if (tap < 2 * NK && more) stage_b(cc + 1, (cc + 1) & 1, tap >> 1, tap & 1);
      } else if (tap == 8 && more) {                     // every wave is past its last read of this group's tile
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
          for (int s = 0; s < 2; ++s) stage_b(cc + 1, 0, k, s);
      }
      STK_H_MFMAS
    }
  }
#undef STK_H_MFMAS
#undef STK_H_FRAGS

  const float unscale = 1.f / (sw * x2::pow2_scale_of(xmax.reduce(red_amax, tid)));
  ep.stage(lds, tid);
  ep.init(p, 0, 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn0 + j * 32 + fc;
    ep.col(p, n);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] *= unscale;
      ep.strip(p, m0 + wm0 + i * 32 + 4 * fk, M, true, n, acc[i][j]);
    }
  }
  ep.finish(p, lds, tid);
}

// shapes of the halo kernel: 3x3 on 16-, 32- or 64-wide maps made of whole 128-pixel tiles, no K split.  (Row strips for the 128- /
// 256-wide maps and a two-buffer form were measured slower in rounds 3-4 and retired in round 6: DESIGN.md "Retired".)
inline int halo_cols(int W) { return (W == 16 || W == 32 || W == 64) ? W : 0; }
inline bool halo_ok(const ConvP& p, int taps, int splits) {
  return taps == 9 && splits == 1 && halo_cols(p.W) != 0 && p.HW % 128 == 0 && p.H * p.W == p.HW;
}

}  // namespace x2d
