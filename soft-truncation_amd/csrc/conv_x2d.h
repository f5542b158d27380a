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
  const float sx = x2::pow2_scale_of(x2::block_amax(xpart, nxpart, reinterpret_cast<float*>(lds)));
  const float sw = x2::weight_scale(q.wp);
  const float unscale = 1.f / (sw * sx);
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
// Round 4: W = 128 instantiates ROW STRIPS for the wide maps of the 256 x 256 net (map width 128 or 256): a tile is 128
// consecutive pixels of ONE row starting at column x0 = hw0 % width, its halo tile 3 rows x 130 columns (390 rows of 64 bytes per
// plane: 50 instead of 144 B-side DMA instructions per channel group and workgroup; 67 KB of LDS, two workgroups per CU).
// Bit-identical to gemm_kernel, but slower than it (see halo_cols): kept behind STK_X2D_HALO_WIDE=1.
// NB = 2: two B buffers, the next group's tile streams in one piece per wave and chunk (two workgroups per CU);
// NB = 1: one B buffer, refilled in a burst behind the last tap's fragment reads (three workgroups per CU cover the wait).
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
  const float sx = x2::pow2_scale_of(x2::block_amax(xpart, nxpart, reinterpret_cast<float*>(lds)));
  const float sw = x2::weight_scale(q.wp);
  const float unscale = 1.f / (sw * sx);
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

// ---- small tiles for small problems (round 5) ------------------------------------------------------------------------
// Where 128 x 128 tiles do not fill the chip (8x8 / 4x4 maps at batch 128: 128 / 32 tiles; every level below 64x64 of the
// 256x256 net at batch 4) gemm_kernel splits K, writes partial [M][N] slabs and a second launch sums them: two launches with
// their prologues per layer and direction, 20-47 us for 0.1-10 GFLOP (profiles/r04_ksplit_sweep.txt, r05_ksplit_b4.txt: no
// choice of split moves the floor of ~16 us).  Here the tile is 64 output channels x 64 pixels -- four times the tiles, so the
// 8x8 level at batch 128 (512 tiles) and the 32x32 level at batch 4 (256) need no split at all, the 4x4 level two to four
// channel-group splits instead of six -- with the halo staging of gemm_halo_kernel adapted to it:
//   * a tile is 64 consecutive pixels of one image (W >= 8: 64 / W whole rows) or FOUR whole 4x4 images, each with its own
//     6 x 6 halo tile; the halo tile of a 32-channel group is staged once ((IMGS (R + 2) (W + 2) rows of 64 bytes per plane: 7
//     or 9 DMA instructions) into the other of two B buffers, one piece per wave and chunk, while the current group computes;
//   * the weights of a chunk are 64 rows: ONE DMA instruction per wave and plane, in a ring of three buffers filled two
//     chunks ahead (a chunk is 6 MFMAs per wave here, far less than a DMA's latency), one barrier per chunk;
//   * the first loads are issued BEFORE the scale records are reduced (only the epilogue needs the scales).
// A wave owns a 32 x 32 sub-tile: 8 fragment reads per 6 MFMAs -- the kernel is LDS-read-bound, not matrix-bound, which is
// the price of the parallelism; K is split over whole channel groups only (EpSlab + slab sum, as before).
template <int W, class EP>
__global__ __launch_bounds__(256, 2) void gemm_halo64_kernel(ConvP p, x3::Src q, int M, int Nn, int tiles_m, int tiles_n,
                                                             int ngroups, int groups_per_split,
                                                             const float* __restrict__ xpart, int nxpart) {
  constexpr int TM = 64, TN = 64;
  constexpr int IMGS = W == 4 ? 4 : 1;                                    // whole 4x4 images per tile
  constexpr int R = W == 4 ? 4 : TN / W;                                  // map rows of one image in the tile
  constexpr int TP = W + 2, TIMG = (R + 2) * TP;                          // halo rows (of 64 bytes) per image
  constexpr int HR = IMGS * TIMG, NI = (HR + 15) / 16, HRP = NI * 16, NK = (NI + 3) / 4;
  constexpr int NS = 3;                                                   // ring of weight tiles
  constexpr int A_PLANE = TM * 64, A_BYTES = 2 * A_PLANE;                 // 8 KB per chunk
  constexpr int B_PLANE = HRP * 64, B_BYTES = 2 * B_PLANE;
  static_assert(2 * NK <= 8, "the B pieces of a group are issued one per chunk over taps 0..7");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * A_BYTES + 2 * B_BYTES];
  __shared__ float red[4];
  unsigned char* const As = lds;
  unsigned char* const Bs = lds + NS * A_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = id % ntiles, zs = id / ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * TM, n0 = tn * TN;
  const int g_begin = zs * groups_per_split, g_end = min(ngroups, g_begin + groups_per_split);
  const int c_end = g_end * 9;

  const unsigned seg_src = (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
  const unsigned a_plane2 = (unsigned)q.taps * q.Mpad * q.Kc * 2u;
  const unsigned a_chunk2 = (unsigned)q.Mpad * KC * 2u;
  const __amdgpu_buffer_rsrc_t a_rs = x3::make_rsrc(reinterpret_cast<const unsigned char*>(q.wp) + x2::HEADER, 2L * a_plane2);
  const unsigned a_voff = (unsigned)(m0 + 16 * wid + (lane >> 2)) * 64u + seg_src;     // wave w stages rows 16 w .. 16 w + 15
  const __amdgpu_buffer_rsrc_t b_rs = x3::make_rsrc(q.pl, 2L * q.pl_stride);
  const unsigned b_ps = (unsigned)q.pl_stride;
  // halo rows of this lane: instruction i = wid + 4 k covers LDS rows 16 i .. 16 i + 15, lane -> row 16 i + lane / 4
  unsigned b_voff[NK];
  {
    const int Cb = q.Kc >> 5;
    const int b0 = n0 / p.HW, hw0 = n0 - b0 * p.HW, y0 = W == 4 ? 0 : hw0 / p.W;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int row = 16 * (wid + 4 * k) + (lane >> 2);
      const int img = row / TIMG, rr = row - img * TIMG;
      const int ty = rr / TP, tx = rr - ty * TP;
      const int b = b0 + img, y = y0 - 1 + ty, x = tx - 1;
      const bool ok = row < HR && b < p.N && y >= 0 && y < p.H && x >= 0 && x < p.W;
      b_voff[k] = ok ? ((unsigned)(b * Cb) * p.HW + (unsigned)(y * p.W + x)) * 64u + seg_src : 0x80000000u;
    }
  }
  auto stage_a = [&](int c, int slot) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      dma16(a_rs, As + slot * A_BYTES + s * A_PLANE + (16 * wid) * 64, a_voff, (unsigned)c * a_chunk2 + s * a_plane2);
  };
  auto stage_b = [&](int cc, int buf, int k, int s) {                     // k, s compile-time at every call site
    if (wid + 4 * k >= NI) return;                                        // wave-uniform
    dma16(b_rs, Bs + buf * B_BYTES + s * B_PLANE + (wid + 4 * k) * 1024, b_voff[k],
          (unsigned)cc * (unsigned)p.HW * 64u + s * b_ps);
  };

  // first loads: the whole halo tile of the first group, the weights of the first two chunks
  int c = g_begin * 9;
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int s = 0; s < 2; ++s) stage_b(g_begin, 0, k, s);
  stage_a(c, 0);
  if (c + 1 < c_end) stage_a(c + 1, 1);

  EP ep;
  ep.preload(p, m0, n0, TN, M, Nn, tid);
  // the scale records: loaded now (older than every load of the main loop, so they never hold up its counted waits), reduced
  // after it -- only the epilogue needs them
  float pmax = xpart[tid];
  if (nxpart > x2::NPART) pmax = fmaxf(pmax, xpart[x2::NPART + tid]);
  const float sw = x2::weight_scale(q.wp);

  floatx16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  const int wm0 = (wid & 1) * 32, wn0 = (wid >> 1) * 32;
  const int fk = lane >> 5, fc = lane & 31;
  const int fsw = (fc >> 2) & 3;
  const unsigned char* a_rd = As + (wm0 + fc) * 64;
  const int ka0 = ((0 + fk) ^ fsw) * 16, ka1 = ((2 + fk) ^ fsw) * 16;
  int b_row;                                                              // halo row of this lane's pixel for the centre tap
  {
    const int pt = wn0 + fc;
    if (W == 4) b_row = (pt >> 4) * TIMG + (((pt & 15) >> 2) + 1) * TP + (pt & 3) + 1;
    else b_row = (pt / W + 1) * TP + (pt % W) + 1;
  }
  halfx8 a[2], b[2];
#define STK_S_MFMAS                                                                                         \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);                                    \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);                                    \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
#define STK_S_FRAGS(KA, KSEG)                                                                               \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                           \
    a[s] = *reinterpret_cast<const halfx8*>(abase + s * A_PLANE + (KA));                                    \
    b[s] = *reinterpret_cast<const halfx8*>(bbase + s * B_PLANE + brow * 64 + ((((KSEG) + fk) ^ (brow >> 2)) & 3) * 16); \
  }

  int slot = 0;
  for (int cc = g_begin; cc < g_end; ++cc) {
    const int bbuf = (cc - g_begin) & 1;
    const unsigned char* bbase = Bs + bbuf * B_BYTES;
    const bool more = cc + 1 < g_end;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++c) {
      const int brow = b_row + (tap / 3 - 1) * TP + (tap % 3 - 1);
      // this wave's weights of chunk c have landed (issued after them: at most one B piece and the two of chunk c + 1), and
      // with them every B piece issued before -- all of this group's halo tile
      if (c + 1 < c_end) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // ... everybody's; and nobody reads chunk c - 1 / group cc - 1 any more
      if (tap < 2 * NK && more) stage_b(cc + 1, bbuf ^ 1, tap >> 1, tap & 1);
      if (c + 2 < c_end) stage_a(c + 2, slot >= 1 ? slot - 1 : NS - 1);      // (slot + 2) % 3
      const unsigned char* abase = a_rd + slot * A_BYTES;
      STK_S_FRAGS(ka0, 0)
      STK_S_MFMAS
      STK_S_FRAGS(ka1, 2)
      STK_S_MFMAS
      slot = slot == NS - 1 ? 0 : slot + 1;
    }
  }
#undef STK_S_MFMAS
#undef STK_S_FRAGS

  float unscale;
  {
    const float m = wave_max(pmax);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    unscale = 1.f / (sw * x2::pow2_scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
  }
  ep.stage(lds, tid);
  ep.init(p, 0, zs);
  {
    const int n = n0 + wn0 + fc;
    const bool nok = n < Nn;
    ep.col(p, nok ? n : 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] *= unscale;
    ep.strip(p, m0 + wm0 + 4 * fk, M, nok, nok ? n : 0, acc);
  }
  ep.finish(p, lds, tid);
}

// The small-tile plan of a plane-operand 3x3 call (STK_X2D_T64=0: off): shapes gemm_halo64_kernel takes and for which 128 x 128
// tiles would not fill the chip.  K is split over whole 32-channel groups, and only while the tiles alone leave CUs idle.
struct T64Plan { int ok; int splits; int groups_per_split; };
// OFF by default (STK_X2D_T64=1 switches it on): the kernel-level gains below did not survive inside the training step.  Measured with
// bench.py's event brackets on the eager steps (profiles/r05_t64_in_situ.txt): the 8x8 layers at batch 128 take 66.5 us on the small
// tiles against 43.1 us in the back-to-back micro-benchmark -- and 44 us on the 128-tile K-split form, which measures the same in both
// settings; the step went 39.3-39.5 -> 40.0 ms (CIFAR-10 net), 39.64 -> 39.62 (256x256 net, batch 4), 124.0 -> 124.8 (64x64 net).  A
// kernel that is bound by LDS reads and barriers rather than by the matrix pipe runs at whatever clock its neighbours leave: alone it
// enjoys the boost clock, inside a step of power-limited GEMMs it does not, while the matrix-bound kernels are power-limited either way.
inline int t64_mode() { static const int v = [] { const char* e = getenv("STK_X2D_T64"); return e ? atoi(e) : 0; }(); return v; }
inline T64Plan t64_plan(const ConvP& p, int taps, int Kc, int M, long Ng) {
  T64Plan r = {0, 1, 0};
  if (!t64_mode() || taps != 9 || p.H * p.W != p.HW || p.stride != 1) return r;
  const bool geom = (p.W == 4 && p.H == 4) || ((p.W == 8 || p.W == 16 || p.W == 32) && p.HW % 64 == 0);
  if (!geom || Kc % 32) return r;
  const long t128 = (long)stk_cdiv(M, 128) * stk_cdiv(Ng, 128);
  if (t128 >= 192) return r;                              // the large tiles fill the chip: their kernels are the faster ones
  const long tiles = (long)stk_cdiv(M, 64) * stk_cdiv(Ng, 64);
  const int ngroups = Kc / 32;
  // STK_T64_WGS: workgroups worth splitting K for (default 512 = two per CU); at least two channel groups (18 chunks) per split
  static const long target = [] { const char* e = getenv("STK_T64_WGS"); return e && atol(e) > 0 ? atol(e) : 512L; }();
  long splits = tiles >= target ? 1 : stk_cdiv(target, tiles);
  if (splits > ngroups / 2) splits = ngroups / 2;
  if (splits < 1) splits = 1;
  r.groups_per_split = (int)stk_cdiv((long)ngroups, splits);
  r.splits = stk_cdiv(ngroups, r.groups_per_split);
  // Per unit of work the small tiles are about half as efficient as the large ones (8 fragment reads per 6 MFMAs, a barrier per
  // 6 MFMAs): they win by what they save -- the second launch, the slabs, idle CUs -- only while a workgroup's share of K is
  // short.  Measured (profiles/r05_t64_ab.txt, us, 128-tile K-split form -> small tiles): 256 -> 256 at 8x8, batch 128 (72 chunks, no
  // split) 47.3 -> 43.1; at 4x4 (18 chunks) 24.3 -> 20.0; 512 -> 256 at 4x4 (36) 34.0 -> 28.9; 256 -> 256 at 16x16, batch 16 (36)
  // 33.9 -> 26.4; but 512 -> 256 at 8x8, batch 128 (144 chunks) 69.2 -> 99.3 and at 16x16, batch 16 (72 chunks + slabs) 44.6 -> 53.8.
  // STK_X2D_T64=2 lifts the limit (A/B).
  const int chunks = r.groups_per_split * 9;
  if (t64_mode() != 2 && chunks > (r.splits == 1 ? 72 : 36)) return r;
  r.ok = 1;
  return r;
}

// shapes of the halo kernel: 3x3 on 16- or 32-wide maps made of whole 128-pixel tiles, no K split (STK_X2D_HALO=0: off)
inline int halo_mode() { static const int v = [] { const char* e = getenv("STK_X2D_HALO"); return e ? atoi(e) : 1; }(); return v; }
// template width of the kernel that takes a map of width W (0 = none): the map width itself up to 64, row strips of 128 beyond
inline int halo_cols(int W) {
  // row strips on the 128- / 256-wide maps: measured SLOWER than x2d::gemm_kernel (256 x 256 net, batch 4: forward 187.9 -> 203.1 us,
  // data gradient 184.0 -> 190.9: 67 KB of LDS leave two workgroups per CU for the refill burst) -- off unless STK_X2D_HALO_WIDE=1
  static const bool wide = [] { const char* e = getenv("STK_X2D_HALO_WIDE"); return e && atoi(e) != 0; }();
  if (W == 16 || W == 32) return W;
  if (W == 64 && halo_mode() == 1) return 64;
  if ((W == 128 || W == 256) && halo_mode() == 1 && wide) return 128;
  return 0;
}
inline bool halo_ok(const ConvP& p, int taps, int splits) {
  return halo_mode() != 0 && taps == 9 && splits == 1 && halo_cols(p.W) != 0 && p.HW % 128 == 0 && p.H * p.W == p.HW;
}

}  // namespace x2d
