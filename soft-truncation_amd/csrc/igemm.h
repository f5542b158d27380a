// igemm.h -- the fp32 MFMA GEMM core shared by the convolution (fwd / dgrad / wgrad) and the
// batched strided GEMM entry points of libstk (gfx950).
//
// Why fp32 MFMA: the reference computes in fp32 and parity is judged against its CPU path, so the
// contraction runs on v_mfma_f32_32x32x2_f32 -- exact fp32 (bit-equal to an fmaf chain in k order),
// 64 FLOP/clk/SIMD = 157.3 TFLOP/s on MI355X, ~2.4x what an f32 VALU GEMM reaches.  gfx950 has no
// TF32/xf32 path, so there is no cheaper fp32-input instruction to pick.
//
// Structure (one 256-thread workgroup = 4 waves, one output tile BM x BN):
//   * A tile [KC][BM] and B tile [KC][BN] live in LDS, k-major with an odd row pitch (BM+1 / BN+1):
//     MFMA operand reads (lane l -> row 2*ks + (l>>5), column base + (l&31)) are 32 consecutive floats
//     per half-wave -> conflict-free ds_read_b32, and the staging stores are conflict-free for both
//     lane-along-M/N and lane-along-K producers.
//   * Each wave owns a (BM/2) x (BN/2) quadrant as TM x TN tiles of 32x32, accumulators in registers
//     (64 VGPRs for the 128x128 tile).  Per k-step of 2 a wave issues TM+TN ds_read_b32 and TM*TN
//     MFMAs (64 cycles each), so LDS and global traffic sit far below the matrix pipe's time.
//   * Global -> register prefetch of chunk c+1 is issued before the MFMAs of chunk c and written to LDS
//     after them (issue-early / write-late), so HBM/L2 latency hides under >= 1.1k cycles of MFMA.
//   * Operands are produced by "loader" functors straight from the tensors in HBM: im2col, the
//     channel-concat of two sources, weight layouts ([Cout,Cin,KH,KW] and NIN's [Cin,Cout]) and
//     zero padding are all address arithmetic in the loader -- no im2col buffer, no concat buffer, no
//     weight transform pass ever touches HBM.
//   * (A double-buffered variant with one barrier per chunk and the LDS writes placed between the MFMAs was
//     measured 5-10% SLOWER: 74 KB of LDS halves the resident workgroups per CU, and three or four
//     independent workgroups per CU hide the staging phase better than one deeper pipeline does.)
//   * Epilogue functors fuse bias, the time-embedding add, the residual add, the 1/sqrt(2) rescale and
//     beta-accumulation into the accumulator write-out, which is coalesced along N (lanes = columns).
//   * blockIdx is remapped XCD-aware so tiles sharing an activation panel stay on one L2.
#pragma once
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace igemm {

// Tile configuration: BM x BN block tile, KC k-chunk (even), 4 waves in a 2 x 2 arrangement.
template <int BM_, int BN_, int KC_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, KC = KC_;
  static constexpr int WM = BM / 2, WN = BN / 2;      // per-wave quadrant
  static constexpr int TM = WM / 32, TN = WN / 32;    // 32x32 MFMA tiles per wave
  static constexpr int LDA = BM + 1, LDB = BN + 1;
  static constexpr int NA = BM * KC / 256, NB = BN * KC / 256;   // staged elements per thread
  static constexpr int LDS_FLOATS = KC * (LDA + LDB);
  static_assert(KC % 2 == 0 && (BM * KC) % 256 == 0 && (BN * KC) % 256 == 0, "bad tile");
};

// ------------------------------------------------------------------------------------------------
// Thread -> element mappings for staging a [KC][B] tile (B = BM or BN).
//   MnMajor: lanes walk the M/N index (contiguous in memory for that operand); thread handles a
//            fixed mn and KC/(256/B) k-rows:  mn = tid % B,  kk = (tid / B) * (KC / (256/B)) + r.
//   KMajor : lanes walk k:  flat e = tid + 256 r,  mn = e / KC,  kk = e % KC.
// Both give NA = B*KC/256 elements per thread.
// ------------------------------------------------------------------------------------------------
template <int B, int KC>
struct MnMajor {
  static constexpr int GROUPS = 256 / B;          // thread groups along k
  static constexpr int PER = KC / GROUPS;         // k-rows per thread
  static_assert(256 % B == 0 && KC % GROUPS == 0, "bad MnMajor tile");
  __device__ static __forceinline__ int mn(int tid) { return tid % B; }
  __device__ static __forceinline__ int kgroup(int tid) { return tid / B; }
  // k-row of element r (r is a compile-time loop index)
  __device__ static __forceinline__ int kk(int tid, int r) { return kgroup(tid) * PER + r; }
};
template <int B, int KC>
struct KMajor {
  static constexpr int PER = B * KC / 256;
  __device__ static __forceinline__ int mn(int tid, int r) { return (tid + 256 * r) / KC; }
  __device__ static __forceinline__ int kk(int tid, int r) { return (tid + 256 * r) % KC; }
};

// ------------------------------------------------------------------------------------------------
// The kernel.  AL / BL: loaders with
//     __device__ void init(const P&, int tile_origin, int tid, int zb);
//     __device__ void load(const P&, int k0, float (&r)[N]);   // global -> registers: UNCONDITIONAL loads from
//                                                              //   safe addresses + a validity bitmask kept aside
//     __device__ void store(const float (&r)[N], float* lds);  // registers -> LDS tile [KC][LD], invalid -> 0
//   (the zeroing happens at store time, i.e. after the MFMAs of the previous chunk, so the loads of a chunk
//    are never waited for before the matrix work that hides them)
// EP: epilogue with
//     __device__ void init(const P&, int zb, int zs);
//     __device__ void col(const P&, int n);                     // per output column setup
//     __device__ void strip(const P&, int mbase, int M, bool nok, int n, const floatx16& acc);
//                                                               // 16 rows mbase + strip_row(e) of column n
// Grid: x = tiles (XCD-remapped, m fastest), y = split-K slice, z = batch;  or, with flat_z > 0, one flat
// x dimension of flat_z * tiles * splits blocks (see the kernel body).
// ------------------------------------------------------------------------------------------------
template <class C, class P, class AL, class BL, class EP>
__global__ __launch_bounds__(256) void kernel(P p, int M, int N, int K, int tiles_m, int tiles_n, int k_per_split,
                                              int flat_z) {
  __shared__ float lds[C::LDS_FLOATS];
  float* As = lds;
  float* Bs = lds + C::KC * C::LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int ntiles = tiles_m * tiles_n;
  int tile, zs, zb;
  if (flat_z > 0) {
    // wgrad: grid.x = flat_z (taps) x tiles x splits with the tap index fastest, so the blocks that read the
    // SAME dy / x panels (one per tap) are neighbours in the XCD-remapped order and share them in one L2
    // instead of fetching them from HBM once per tap.
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    zb = id % flat_z;
    const int rest = id / flat_z;
    tile = rest % ntiles;
    zs = rest / ntiles;
  } else {
    tile = xcd_remap(blockIdx.x, ntiles);
    zs = blockIdx.y;
    zb = blockIdx.z;
  }
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const int k_begin = zs * k_per_split;
  const int k_end = min(K, k_begin + k_per_split);

  AL al; BL bl;
  al.init(p, m0, tid, zb);
  bl.init(p, n0, tid, zb);

  floatx16 acc[C::TM][C::TN];
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float ra[C::NA], rb[C::NB];
  if (k_begin < k_end) {
    al.load(p, k_begin, ra);
    bl.load(p, k_begin, rb);
  }
  const int wm0 = (wid & 1) * C::WM, wn0 = (wid >> 1) * C::WN;
  const int fk = lane >> 5, fc = lane & 31;

  for (int k0 = k_begin; k0 < k_end; k0 += C::KC) {
    al.store(ra, As);
    bl.store(rb, Bs);
    __syncthreads();
    if (k0 + C::KC < k_end) {          // prefetch the next chunk; consumed after the MFMAs below
      al.load(p, k0 + C::KC, ra);
      bl.load(p, k0 + C::KC, rb);
    }
#pragma unroll
    for (int ks = 0; ks < C::KC / 2; ++ks) {
      float a[C::TM], b[C::TN];
      const float* arow = As + (2 * ks + fk) * C::LDA + wm0 + fc;
      const float* brow = Bs + (2 * ks + fk) * C::LDB + wn0 + fc;
#pragma unroll
      for (int i = 0; i < C::TM; ++i) a[i] = arow[i * 32];
#pragma unroll
      for (int j = 0; j < C::TN; ++j) b[j] = brow[j * 32];
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // The epilogue handles one 16-register strip (one column, 16 rows) at a time so that it can batch its
  // own loads (bias / residual / old values) instead of waiting on them one by one.
  EP ep;
  ep.init(p, zb, zs);
#pragma unroll
  for (int j = 0; j < C::TN; ++j) {
    const int n = n0 + wn0 + j * 32 + fc;
    const bool nok = n < N;
    ep.col(p, nok ? n : 0);
#pragma unroll
    for (int i = 0; i < C::TM; ++i) ep.strip(p, m0 + wm0 + i * 32 + 4 * fk, M, nok, nok ? n : 0, acc[i][j]);
  }
}

// row of strip element e (relative to the strip base)
__device__ __forceinline__ constexpr int strip_row(int e) { return (e & 3) + 8 * (e >> 2); }

// Zero a value unless bit `idx` of `okm` is set -- two VALU ops (bfe + and), no VCC, no branch.
__device__ __forceinline__ float keep_if(float v, unsigned okm, int idx) {
  const int m = -(int)((okm >> idx) & 1u);
  return __int_as_float(__float_as_int(v) & m);
}

// Generic LDS store helpers -----------------------------------------------------------------------------
template <int B, int KC, int LD>
__device__ __forceinline__ void store_mn_major(const float (&r)[B * KC / 256], float* t, int tid) {
  using Mp = MnMajor<B, KC>;
  const int mn = Mp::mn(tid), kg = Mp::kgroup(tid) * Mp::PER;
#pragma unroll
  for (int i = 0; i < Mp::PER; ++i) t[(kg + i) * LD + mn] = r[i];
}
template <int B, int KC, int LD>
__device__ __forceinline__ void store_k_major(const float (&r)[B * KC / 256], float* t, int tid) {
  using Mp = KMajor<B, KC>;
#pragma unroll
  for (int i = 0; i < Mp::PER; ++i) t[Mp::kk(tid, i) * LD + Mp::mn(tid, i)] = r[i];
}

}  // namespace igemm
