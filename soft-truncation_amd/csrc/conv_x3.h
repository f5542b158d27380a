// conv_x3.h -- 3x3 / stride-1 / pad-1 convolution (forward and data-gradient) on the bf16 matrix pipe with a
// three-way operand split ("x3"), for gfx950.  Included by conv.hip inside its anonymous namespace.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate (157 TFLOP/s); the bf16 MFMA is 16x faster.  An fp32
// value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand bits, taken by truncation):
//     a = a0 + a1 + a2,   b = b0 + b1 + b2
//     a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + O(2^-24 |ab|)
// so six bf16 MFMAs (products exact, fp32 accumulate) give the fp32 product to fp32 rounding accuracy -- measured
// against the double-precision oracle the error is the same as (slightly below) the f32-input MFMA's, see
// tests/test_gpu_kernels.py -- at 6/16 of the matrix-pipe time.
//
// GEMM view (both directions share the kernel):   out[m, n] = sum_k  Wp[m, k] * act[k, n]
//     forward  m = co, n = (b, y, x), k = (tap, ci):   act = x[b, ci, y + kh - 1, x + kw - 1]
//     dgrad    m = ci, n = (b, y, x), k = (tap', co):  act = dy[b, co, y + kh' - 1, x + kw' - 1], tap' = 8 - tap
// K is ordered (32-channel group, tap, channel in group): a 32-wide k chunk is one tap and 32 consecutive channels,
// so the tap (hence the halo test and the pixel shift) is uniform over the chunk and a thread's 16 loads differ only
// by a scalar channel-plane offset; and the nine chunks of a channel group follow each other, so the nine shifted
// reads of the same 32 x (128 + halo) activation patch hit L1 / L2 (with the taps outermost they were 9 sweeps over
// all channels and measured 3-6x the algorithmic bytes at the memory side).
//
//   * weights are re-laid out and split ONCE per call by wprep_kernel into Wp[split][k / 32][tap][row (padded to 128)][k % 32]
//     bf16 (<= 14 MB, L2 resident): the A loader is six 16-byte copies per thread and chunk, no conversion;
//   * activations are split in the B loader when they are written to LDS (and/sub/perm, ~6 VALU per element);
//   * LDS tiles are [split][row][32 k] bf16 with an 80-byte row pitch: the MFMA operand reads (one ds_read_b128
//     per lane = 8 consecutive k of one row) and the staging writes are bank-conflict free;
//   * 128 x 128 tile, 4 waves of 64 x 64 (2 x 2 MFMA tiles): per 16-k step 12 ds_read_b128 and 24 MFMAs per wave,
//     small terms first so the accumulation order is fixed.
//
// Weight gradient (same kernel, other loaders):  dw[tap][co, ci] = sum_pixels dy[co, px] * x[ci, px + tap shift]:
// both operands have k = pixels contiguous in NCHW, so a thread stages 16 consecutive pixels of one channel (the x
// operand shifted by the tap, halo masked); one GEMM per tap, split over K into partial slabs that
// splitk_reduce_kernel sums in a fixed order (no float atomics).
#pragma once

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace x3 {

constexpr int KC = 32;
constexpr int PITCH = 80;              // bytes per LDS row: 32 bf16 + 16 bytes of padding
constexpr int PLANE = 128 * PITCH;     // one split plane of one operand
constexpr int OPER = 3 * PLANE;
constexpr int LDS_BYTES = 2 * OPER;    // 61440

struct Src {             // the activation operand: channel-concat of two NCHW tensors
  const float* s1; const float* s2; int S1, S2;
  const unsigned short* wp; int Mpad; int Kc;     // prepared weights, padded row count, channels (= S1 + S2)
  int taps;                                       // 9 (3x3, pad 1) or 1 (1x1)
  const unsigned char* pl; long pl_stride;        // pre-split activation planes (conv_pl.h) and bytes per plane, or null
};

__device__ __forceinline__ float hi_part(float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); }
// pack the bf16 (upper) halves of two floats: low half <- lo, high half <- hi
__device__ __forceinline__ unsigned pack_hi(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}

// Wp[split][k / 32][tap][row][k % 32] = split_s( W(row, k, tap) ), rows >= R are zero: one (channel group, tap) is a
// dense [Mpad][32] bf16 block, i.e. the A tile of one chunk is 8 KB of consecutive memory per plane.
// W(row, k, tap) = w[row*sm + k*sk + (flip ? taps-1-tap : tap)]:
//     forward  [Cout,Cin,kh,kw]: sm = Cin*taps, sk = taps      NIN [Cin,Cout]: sm = 1, sk = Cout
//     dgrad    [Cout,Cin,kh,kw]: sm = taps, sk = Cin*taps, flip   NIN: sm = Cout, sk = 1
// One thread per (row, k): reads its taps (contiguous).
struct WprepDesc {        // == StkWprepDesc (include/stk.h)
  const float* w; unsigned short* wp; long sm, sk; int M, Kc, Mpad, taps, flip, reserved;
};
__device__ __forceinline__ void wprep_one(const float* __restrict__ w, unsigned short* __restrict__ out, int R, int Kd,
                                          int Mpad, long sm, long sk, int taps, int flip, long i) {
  const long total = (long)Mpad * Kd;
  if (i >= total) return;
  // k fastest inside a group of 32 so that the 2-byte stores of a wave are contiguous
  const int kl = (int)(i & 31);
  const long rest = i >> 5;
  const int row = (int)(rest % Mpad), cc = (int)(rest / Mpad), k = cc * 32 + kl;
  const long plane = (long)taps * Mpad * Kd;
  const float* s = w + (row < R ? row * sm + k * sk : 0);
  for (int t = 0; t < taps; ++t) {
    const float a = row < R ? s[flip ? taps - 1 - t : t] : 0.f;
    const float h0 = hi_part(a), r1 = a - h0;
    const float h1 = hi_part(r1), h2 = r1 - h1;           // h2 has <= 8 significant bits: exact in bf16
    const long o = (((long)cc * taps + t) * Mpad + row) * 32 + kl;
    out[o] = (unsigned short)(__float_as_uint(h0) >> 16);
    out[plane + o] = (unsigned short)(__float_as_uint(h1) >> 16);
    out[2 * plane + o] = (unsigned short)(__float_as_uint(h2) >> 16);
  }
}
__global__ __launch_bounds__(256) void wprep_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int R,
                                                    int Kd, int Mpad, long sm, long sk, int taps, int flip) {
  wprep_one(w, out, R, Kd, Mpad, sm, sk, taps, flip, (long)blockIdx.x * 256 + threadIdx.x);
}
// All layers of a network in one launch: blockIdx.y picks the descriptor, blockIdx.x walks its (row, k) pairs
// (grid.x is sized for the largest layer; blocks past the end of a smaller one exit).
__global__ __launch_bounds__(256) void wprep_batch_kernel(const WprepDesc* __restrict__ descs) {
  const WprepDesc d = descs[blockIdx.y];
  wprep_one(d.w, d.wp, d.M, d.Kc, d.Mpad, d.sm, d.sk, d.taps, d.flip, (long)blockIdx.x * 256 + threadIdx.x);
}

// Staging is cut into 24 slices so that the kernel can place one slice after each MFMA of a 24-MFMA group (the
// hipcc scheduler, left alone, runs the whole staging phase first and the MFMAs after it).  A loader provides
//     st(g, tile)        slice g of "registers of the current chunk -> LDS"
//     ld(g, p, q, c)     slice g of "global -> registers for chunk c"
// and st(g) of a chunk always precedes ld(g) of the next one, so a register is reloaded only after its slice
// consumed it.
//
// Splitter shared by the converting loaders: 16 fp32 values of one LDS row (16 consecutive k at k offset k0) ->
// three bf16 planes.  Slices 0..7 convert one pair each (invalid elements zeroed by mask bit), slices 8..13 write
// one 16-byte piece each (2 x ds_write_b128 per plane).
struct Split16 {
  unsigned pk[3][8];
  __device__ __forceinline__ void st(int g, const float (&r)[16], unsigned okm, unsigned char* tile, int row, int k0) {
    if (g < 8) {
      const float v0 = igemm::keep_if(r[2 * g], okm, 2 * g), v1 = igemm::keep_if(r[2 * g + 1], okm, 2 * g + 1);
      const float r0 = v0 - hi_part(v0), r1 = v1 - hi_part(v1);
      const float t0 = r0 - hi_part(r0), t1 = r1 - hi_part(r1);
      pk[0][g] = pack_hi(v0, v1);
      pk[1][g] = pack_hi(r0, r1);
      pk[2][g] = pack_hi(t0, t1);
    } else if (g < 14) {
      const int s = (g - 8) >> 1, h = (g - 8) & 1;
      *reinterpret_cast<u32x4*>(tile + s * PLANE + row * PITCH + k0 * 2 + h * 16) =
          u32x4{pk[s][4 * h], pk[s][4 * h + 1], pk[s][4 * h + 2], pk[s][4 * h + 3]};
    }
  }
};

// ---- loaders: init(p, q, tile origin, tid, zb) / load(p, q, chunk) / store(LDS operand tile) ------------------
// All global reads are raw BUFFER loads: a wave-uniform resource (tensor base + size in SGPRs), one 32-bit
// per-lane byte offset and a scalar offset per load, so a chunk's loads cost no per-load address VALU; and a read
// outside the tensor returns 0 instead of faulting, which is how halo lanes are zeroed (offset bit 31 set) and why
// a shifted 16-pixel run may start one element before / end one element after its tensor.
// (Buffer offsets are 32-bit: the host side keeps these kernels to tensors below 2 GB.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// forward / dgrad A: prepared weights, six 16-byte pieces per chunk (split j>>1, row (tid>>2) + 64 (j&1), segment tid&3)
struct WpLoader {
  __amdgpu_buffer_rsrc_t rs; unsigned voff, plane2, chunk2; int row, seg;
  u32x4 r[6];
  __device__ __forceinline__ void init(const ConvP&, const Src& q, int m0, int tid, int) {
    row = tid >> 2; seg = tid & 3;
    plane2 = (unsigned)q.taps * q.Mpad * q.Kc * 2u;     // bytes per split plane
    chunk2 = (unsigned)q.Mpad * KC * 2u;                // bytes per (channel group, tap) block = per chunk
    rs = make_rsrc(q.wp, 3L * plane2);
    voff = ((unsigned)(m0 + row) * KC + seg * 8) * 2u;
  }
  __device__ __forceinline__ void ld(int g, const ConvP&, const Src&, int c) {
    if (g >= 6) return;
    r[g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                         rs, (int)voff, (int)((unsigned)c * chunk2 + (g >> 1) * plane2 + (g & 1) * 64u * KC * 2u), 0));
  }
  __device__ __forceinline__ void st(int g, unsigned char* t) {
    if (g < 6) *reinterpret_cast<u32x4*>(t + (g >> 1) * PLANE + (row + 64 * (g & 1)) * PITCH + seg * 16) = r[g];
  }
};

// forward / dgrad B: activations, lanes along pixels; a thread holds 16 channels of one tap-shifted pixel
template <bool DUAL, int TAPS>
struct ActLoader {
  __amdgpu_buffer_rsrc_t rs1, rs2;
  int nl, kg, tb1, tb2; unsigned mask;
  float r[16]; Split16 sp;
  __device__ __forceinline__ void init(const ConvP& p, const Src& q, int n0, int tid, int) {
    nl = tid & 127;
    kg = __builtin_amdgcn_readfirstlane(tid >> 7);      // which 16 of the chunk's 32 channels
    rs1 = make_rsrc(q.s1, (long)p.N * q.S1 * p.HW * 4);
    rs2 = make_rsrc(q.s2, (long)p.N * (DUAL ? q.S2 : q.S1) * p.HW * 4);
    mask = 0; tb1 = 0; tb2 = 0;
    const int n = n0 + nl;
    if (n < p.N * p.HW) {
      const int b = n / p.HW, hw = n - b * p.HW;
      const int y = hw / p.W, x = hw - y * p.W;
      if (TAPS == 1) mask = 1u;
#pragma unroll
      for (int t = 0; t < (TAPS == 9 ? 9 : 0); ++t) {
        const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1u << t;
      }
      tb1 = b * q.S1 * p.HW + hw;
      tb2 = b * q.S2 * p.HW + hw;
    }
  }
  // slices 8..23 load one channel each (its register was consumed by conversion slice (g - 8) / 2 <= 7)
  __device__ __forceinline__ void ld(int g, const ConvP& p, const Src& q, int c) {
    if (g < 8) return;
    const int cc = TAPS == 9 ? c / 9 : c, tap = c - cc * TAPS;   // scalar: chunk = (32-channel group, tap), tap fastest
    const int ci0 = cc * KC + kg * 16;
    const bool first = !DUAL || ci0 < q.S1;                      // scalar: a chunk never straddles the two sources
    const __amdgpu_buffer_rsrc_t rs = first ? rs1 : rs2;
    const unsigned so = (unsigned)(first ? ci0 : ci0 - q.S1) * p.HW * 4u;
    // halo / out-of-range lanes: offset bit 31 -> outside the buffer -> the load returns 0 (no select, no branch)
    const unsigned dead = (((mask >> tap) & 1u) ^ 1u) << 31;
    const int shift = TAPS == 9 ? (tap / 3 - 1) * p.W + (tap % 3 - 1) : 0;
    const unsigned vo = (unsigned)(((first ? tb1 : tb2) + shift) * 4) | dead;
    r[g - 8] = bload(rs, vo, so + (unsigned)(g - 8) * p.HW * 4u);
  }
  __device__ __forceinline__ void st(int g, unsigned char* t) { sp.st(g, r, 0xffffu, t, nl, kg * 16); }
};

// wgrad operands: rows = channels, k = pixels (contiguous in NCHW).  Thread (row = tid>>1, half = tid&1) holds the
// 16 consecutive pixels k0 + 16*half .. +15 of its channel, shifted by the tap for the x operand.  Requires W a
// power of two >= 4 and H*W a power of two >= 16; SEG = min(W, 16): a 16-pixel run is 16 / SEG whole row segments.
// With a concat input the first source must hold a multiple of 32 channels (a wave's 32 rows share one tensor).
template <bool SHIFT, bool DUAL, int SEG>
struct RowsLoader {
  static_assert(SEG == 16 || SEG == 8 || SEG == 4, "row segment of a 16-pixel run");
  static constexpr int NSEG = 16 / SEG;
  __amdgpu_buffer_rsrc_t rs;
  int rowoff;             // element offset of this thread's channel plane inside image 0 of its tensor
  int bstride;            // elements between images in this thread's tensor
  int row, half, dy, dx; bool rowok; unsigned okm;
  float r[16]; Split16 sp;
  __device__ __forceinline__ void init(const ConvP& p, const Src&, int o0, int tid, int zb) {
    row = tid >> 1; half = tid & 1; okm = 0;
    dy = SHIFT ? zb / 3 - 1 : 0; dx = SHIFT ? zb % 3 - 1 : 0;     // (a 1x1 layer passes the centre tap, zb = 4)
    const int ch = o0 + row;
    if (SHIFT) {                                   // x: channel of the concat input
      rowok = ch < p.Cin;
      const int c = rowok ? ch : 0;
      const bool first = !DUAL || __builtin_amdgcn_readfirstlane(o0 + (tid >> 6) * 32) < p.C1;   // wave-uniform
      rs = first ? make_rsrc(p.x1, (long)p.N * p.C1 * p.HW * 4) : make_rsrc(p.x2, (long)p.N * p.C2 * p.HW * 4);
      rowoff = (first ? c : (c >= p.C1 ? c - p.C1 : 0)) * p.HW;
      bstride = (first ? p.C1 : p.C2) * p.HW;
    } else {                                       // dy: output channel
      rowok = ch < p.Cout;
      rs = make_rsrc(p.dy, (long)p.N * p.Cout * p.HW * 4);
      rowoff = (rowok ? ch : 0) * p.HW;
      bstride = p.Cout * p.HW;
    }
  }
  // slice 14 computes the next run's mask and offsets, slices 15..19 issue its loads (all 16 registers were
  // consumed by the conversion slices 0..7); the mask in use by those slices is replaced at slice 14.
  unsigned os[NSEG], og[NSEG];      // byte offsets of the first element / of elements 1.. of each row segment
  __device__ __forceinline__ void ld(int g, const ConvP& p, const Src&, int c) {
    if (g == 14) {
      const int k = c * KC + half * 16;              // first pixel of the run (global pixel index)
      const bool kin = rowok && k < p.N * p.HW;      // N*H*W is a multiple of 16: a run is inside or outside as a whole
      const int b = k >> p.ohw_shift, hw = k & (p.HW - 1);
      const int y = hw >> p.ow_shift, x0 = hw & (p.W - 1);
      unsigned m = 0;
      if (!SHIFT) {
        m = 0xffffu;
      } else if (SEG == 16) {                        // one row segment of a row of >= 16 pixels
        const bool yok = (unsigned)(y + dy) < (unsigned)p.H;
        m = yok ? 0xffffu : 0u;
        if (dx < 0 && x0 == 0) m &= ~1u;
        if (dx > 0 && x0 + 16 == p.W) m &= ~0x8000u;
      } else {                                       // NSEG whole rows of SEG pixels
#pragma unroll
        for (int sgi = 0; sgi < NSEG; ++sgi)
          if ((unsigned)(y + sgi + dy) < (unsigned)p.H) m |= ((1u << SEG) - 1u) << (sgi * SEG);
        constexpr unsigned FIRST = SEG == 8 ? 0x0101u : 0x1111u;
        if (dx < 0) m &= ~FIRST;
        if (dx > 0) m &= ~(FIRST << (SEG - 1));
      }
      okm = kin ? m : 0u;
      // The run is contiguous in memory, but its element offsets may be negative where they are masked: column -1
      // of the first row of the tensor (the first element of a segment), or whole leading segments when the row
      // above the image is addressed.  A negative voffset plus an immediate is NOT wrapped back into the buffer by
      // the range check, so each segment is read as two pieces -- its first element and the rest -- each from its
      // own offset clamped at 0 (a clamped piece is entirely masked), immediates only inside a piece.  The offsets
      // are made opaque because hipcc otherwise rewrites max(o + 1, 0) * 4 + imm as max(o, -1) * 4 + (imm + 4),
      // i.e. back into the negative-base form.  A run that ends past the tensor reads 0 there (range check).
      const int o = (kin ? b * bstride + rowoff + hw : 0) + dy * p.W + dx;
#pragma unroll
      for (int sgi = 0; sgi < NSEG; ++sgi) {
        os[sgi] = (unsigned)(max(o + sgi * SEG, 0) * 4);
        og[sgi] = (unsigned)(max(o + sgi * SEG + 1, 0) * 4);
        asm volatile("" : "+v"(os[sgi]), "+v"(og[sgi]));
      }
    } else if (g == 15) {
#pragma unroll
      for (int sgi = 0; sgi < NSEG; ++sgi) r[sgi * SEG] = bload(rs, os[sgi], 0);
    } else if (g >= 16 && g < 20) {
      // the 16 - NSEG remaining elements in four slices: quarter q of every segment's tail
      const int q = g - 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int sgi = j / SEG, e = j % SEG;                     // element e >= 1 of segment sgi
        if (e >= 1 && ((e - 1) * 4) / (SEG - 1) == q) r[j] = bload(rs, og[sgi] + (unsigned)(e - 1) * 4u, 0);
      }
    }
  }
  __device__ __forceinline__ void st(int g, unsigned char* t) { sp.st(g, r, okm, t, row, half * 16); }
};

// ---- the kernel: out tile 128 x 128, 4 waves of 64 x 64, chunks of 32 k ---------------------------------------
// Grid (XCD-remapped): one flat dimension of taps_z x tiles x K-splits blocks, z fastest.  taps_z = 9 for the 3x3
// weight gradient (the nine blocks reading the same dy / x panels are neighbours on one XCD's L2), 1 otherwise.
template <class AL, class BL, class EP, bool HAND>
__global__ __launch_bounds__(256) void gemm_kernel(ConvP p, Src q, int M, int Nn, int tiles_m, int tiles_n,
                                                   int nchunks_total, int chunks_per_split, int taps_z) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  unsigned char* As = lds;
  unsigned char* Bs = lds + OPER;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int ntiles = tiles_m * tiles_n;
  // one flat grid dimension of z-batches (taps) x tiles x K splits, z fastest
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int zb = id % taps_z;
  const int rest = id / taps_z;
  const int tile = rest % ntiles;
  const int zs = rest / ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * 128;
  const int c_begin = zs * chunks_per_split;
  const int c_last = min(nchunks_total, c_begin + chunks_per_split) - 1;     // >= c_begin by construction

  AL al; BL bl;
  al.init(p, q, m0, tid, taps_z == 1 ? 4 : zb);      // 1x1 weight gradient: the (unshifted) centre tap
  bl.init(p, q, n0, tid, taps_z == 1 ? 4 : zb);

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm0 = (wid & 1) * 64, wn0 = (wid >> 1) * 64;
  const int fk = lane >> 5, fc = lane & 31;
  const unsigned char* a_rd = As + (wm0 + fc) * PITCH + fk * 16;
  const unsigned char* b_rd = Bs + (wn0 + fc) * PITCH + fk * 16;

  // Pipeline (LDS single-buffered, two barriers per chunk):
  //   B1: chunk c is in LDS          -> read the kk = 0 fragments, 24 MFMAs, read the kk = 1 fragments
  //   B2: nobody reads LDS any more  -> 24 MFMAs of kk = 1, each followed by ONE SLICE of the staging of chunk c + 1
  //                                     (split + LDS write; its global loads were issued one iteration earlier) and
  //                                     of the global loads of chunk c + 2, pinned in place by sched_barrier.
  // So the conversion VALU, the LDS stores and the load issue run in the shadow of the matrix pipe of the same wave
  // (~6 other instructions fit in the 32 cycles of one MFMA), and a global load has a whole chunk period to land.
#define STK_X3_FRAGS(KK)                                                                               \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int s = 0; s < 3; ++s) {         \
    a[i][s] = *reinterpret_cast<const bf16x8*>(a_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);          \
    b[i][s] = *reinterpret_cast<const bf16x8*>(b_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);          \
  }
  // six products per tile, smallest terms first; the four tiles interleave so that consecutive MFMAs never
  // wait on the same accumulator.  MFMA g of a group: product g / 4, tile (g / 2) & 1, g & 1.
  constexpr int SA[6] = {2, 1, 0, 1, 0, 0}, SB[6] = {0, 1, 2, 0, 1, 0};
#define STK_X3_MFMA(G)                                                                                                \
  acc[((G) >> 1) & 1][(G) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[((G) >> 1) & 1][SA[(G) >> 2]], b[(G) & 1][SB[(G) >> 2]], \
                                                                         acc[((G) >> 1) & 1][(G) & 1], 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 24; ++g) { al.ld(g, p, q, c_begin); bl.ld(g, p, q, c_begin); }
  {
    const int c1 = min(c_begin + 1, c_last);
#pragma unroll
    for (int g = 0; g < 24; ++g) { al.st(g, As); bl.st(g, Bs); al.ld(g, p, q, c1); bl.ld(g, p, q, c1); }
  }
  bf16x8 a[2][3], b[2][3];
  for (int c = c_begin; c < c_last; ++c) {
    __syncthreads();                                   // B1
    STK_X3_FRAGS(0)
#pragma unroll
    for (int g = 0; g < 24; ++g) { STK_X3_MFMA(g) }
    STK_X3_FRAGS(1)
    __syncthreads();                                   // B2
    const int c2 = min(c + 2, c_last);                 // (the last iteration re-loads the last chunk: harmless)
    if (HAND) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 24; ++g) {
        STK_X3_MFMA(g)
        al.st(g, As); bl.st(g, Bs);                    // chunk c + 1: registers -> LDS
        al.ld(g, p, q, c2); bl.ld(g, p, q, c2);        // chunk c + 2: global -> registers
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // Two converting loaders (weight gradient): ~26 VALU per slice do not fit behind one MFMA; pinning them there
      // measured 20-25% slower than handing the scheduler the whole phase with a coarse pipeline hint.
#pragma unroll
      for (int g = 0; g < 24; ++g) { STK_X3_MFMA(g) }
#pragma unroll
      for (int g = 0; g < 24; ++g) al.st(g, As);
#pragma unroll
      for (int g = 0; g < 24; ++g) bl.st(g, Bs);
#pragma unroll
      for (int g = 0; g < 24; ++g) al.ld(g, p, q, c2);
#pragma unroll
      for (int g = 0; g < 24; ++g) bl.ld(g, p, q, c2);
#pragma unroll
      for (int g = 0; g < 24; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);     // VALU
        if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
      }
    }
  }
  __syncthreads();
  STK_X3_FRAGS(0)
#pragma unroll
  for (int g = 0; g < 24; ++g) { STK_X3_MFMA(g) }
  STK_X3_FRAGS(1)
#pragma unroll
  for (int g = 0; g < 24; ++g) { STK_X3_MFMA(g) }
#undef STK_X3_MFMA
#undef STK_X3_FRAGS

  EP ep;
  ep.init(p, zb, zs);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn0 + j * 32 + fc;
    const bool nok = n < Nn;
    ep.col(p, nok ? n : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) ep.strip(p, m0 + wm0 + i * 32 + 4 * fk, M, nok, nok ? n : 0, acc[i][j]);
  }
}

// ---- 3x3 weight gradient, three taps per workgroup ---------------------------------------------------------------
// The per-tap GEMM above re-reads and re-splits dy nine times and x nine times.  Here a workgroup owns one kernel ROW
// (kh) of a 128(co) x 64(ci) block of dw: the three taps kw = 0, 1, 2 read the same x pixels shifted by one, so a
// thread loads its 8-pixel run plus one pixel on each side ONCE, splits those 10 values once and writes the three
// shifted windows to LDS; the dy tile is staged once for all three taps.  Per chunk and thread: 26 loads and
// ~170 VALU for 72 MFMAs per wave (per-tap kernel: 32 loads, ~210 VALU for 48).
//   A tile  [3 planes][128 co][32 px]            as in gemm_kernel (RowsLoader, unshifted)
//   B tiles [3 taps][3 planes][64 ci][32 px]     thread (row = tid >> 2, quarter = tid & 3) owns pixels 8q .. 8q + 7
// Waves: 2 (co halves of 64 rows = 2 MFMA tiles) x 2 (ci halves of 32 = 1 MFMA tile); accumulators acc[2][3 taps].
// Requires W a power of two >= 8 (an 8-pixel run stays inside one image row), H*W a power of two >= 32.
constexpr int W3_BPLANE = 64 * PITCH;            // one plane of one tap of the x operand
constexpr int W3_BTAP = 3 * W3_BPLANE;
constexpr int W3_LDS = OPER + 3 * W3_BTAP;       // 30720 + 46080 = 76800 bytes -> two workgroups per CU

template <bool DUAL>
struct Rows3Loader {
  __amdgpu_buffer_rsrc_t rs;
  int rowoff, bstride, row, quarter, dy; bool rowok;
  unsigned okm;            // bit e + 1: element e (-1 .. 8) of the current run is valid
  float r[10];             // elements -1 .. 8
  float h[3][10];          // their three bf16 parts (upper halves of these fp32 values)
  unsigned om, o0, o8;
  __device__ __forceinline__ void init(const ConvP& p, int n0, int tid, int kh) {
    row = tid >> 2; quarter = tid & 3; dy = kh - 1; okm = 0;
    const int ch = n0 + row;
    rowok = ch < p.Cin;
    const int c = rowok ? ch : 0;
    const bool first = !DUAL || __builtin_amdgcn_readfirstlane(n0 + (tid >> 6) * 16) < p.C1;   // wave-uniform
    rs = first ? make_rsrc(p.x1, (long)p.N * p.C1 * p.HW * 4) : make_rsrc(p.x2, (long)p.N * p.C2 * p.HW * 4);
    rowoff = (first ? c : (c >= p.C1 ? c - p.C1 : 0)) * p.HW;
    bstride = (first ? p.C1 : p.C2) * p.HW;
  }
  // slices 0..4 split two elements each, 5..13 pack + write one (tap, plane) window each
  __device__ __forceinline__ void st(int g, unsigned char* t) {
    if (g < 5) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = 2 * g + u;
        const float v = igemm::keep_if(r[e], okm, e);
        const float r1 = v - hi_part(v);
        h[0][e] = v; h[1][e] = r1; h[2][e] = r1 - hi_part(r1);
      }
    } else if (g < 14) {
      const int tap = (g - 5) / 3, s = (g - 5) % 3;          // window of tap kw: elements kw - 1 .. kw + 6 -> h[kw .. kw + 7]
      unsigned char* d = t + tap * W3_BTAP + s * W3_BPLANE + row * PITCH + quarter * 16;
      *reinterpret_cast<u32x4*>(d) = u32x4{pack_hi(h[s][tap], h[s][tap + 1]), pack_hi(h[s][tap + 2], h[s][tap + 3]),
                                           pack_hi(h[s][tap + 4], h[s][tap + 5]), pack_hi(h[s][tap + 6], h[s][tap + 7])};
    }
  }
  // slice 14: mask and offsets of the next run; 15: the two halo elements; 16, 17: the eight pixels of the run
  __device__ __forceinline__ void ld(int g, const ConvP& p, int c) {
    if (g == 14) {
      const int k = c * KC + quarter * 8;
      const bool kin = rowok && k < p.N * p.HW;
      const int b = k >> p.ohw_shift, hw = k & (p.HW - 1);
      const int y = hw >> p.ow_shift, x0 = hw & (p.W - 1);
      unsigned m = (unsigned)(y + dy) < (unsigned)p.H ? 0x3ffu : 0u;
      if (x0 == 0) m &= ~1u;                       // column -1
      if (x0 + 8 == p.W) m &= ~0x200u;             // column W
      okm = kin ? m : 0u;
      // offsets clamped at 0 and opaque, pieces [-1], [0..7], [8]: see RowsLoader
      const int o = (kin ? b * bstride + rowoff + hw : 0) + dy * p.W;
      om = (unsigned)(max(o - 1, 0) * 4); o0 = (unsigned)(max(o, 0) * 4); o8 = (unsigned)(max(o + 8, 0) * 4);
      asm volatile("" : "+v"(om), "+v"(o0), "+v"(o8));
    } else if (g == 15) {
      r[0] = bload(rs, om, 0);
      r[9] = bload(rs, o8, 0);
    } else if (g == 16 || g == 17) {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[1 + (g - 16) * 4 + j] = bload(rs, o0 + (unsigned)((g - 16) * 4 + j) * 4u, 0);
    }
  }
};

template <bool DUAL>
__global__ __launch_bounds__(256) void wgrad3_kernel(ConvP p, int tiles_m, int tiles_n, int nchunks_total,
                                                     int chunks_per_split) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[W3_LDS];
  unsigned char* As = lds;
  unsigned char* Bs = lds + OPER;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int ntiles = tiles_m * tiles_n;
  const int id = xcd_remap(blockIdx.x, gridDim.x);       // kernel row fastest: the three blocks share dy / x panels
  const int kh = id % 3;
  const int rest = id / 3;
  const int tile = rest % ntiles;
  const int zs = rest / ntiles;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * 128, n0 = tn * 64;
  const int c_begin = zs * chunks_per_split;
  const int c_last = min(nchunks_total, c_begin + chunks_per_split) - 1;

  Src q = {};
  RowsLoader<false, false, 16> al;
  Rows3Loader<DUAL> bl;
  al.init(p, q, m0, tid, 4);
  bl.init(p, n0, tid, kh);

  floatx16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.f;

  const int wm0 = (wid & 1) * 64, wn0 = (wid >> 1) * 32;
  const int fk = lane >> 5, fc = lane & 31;
  const unsigned char* a_rd = As + (wm0 + fc) * PITCH + fk * 16;
  const unsigned char* b_rd = Bs + (wn0 + fc) * PITCH + fk * 16;

#define STK_W3_FRAGS(KK)                                                                                   \
  _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                                           \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
      a[i][s] = *reinterpret_cast<const bf16x8*>(a_rd + s * PLANE + i * 32 * PITCH + (KK) * 32);             \
    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                            \
      b[t][s] = *reinterpret_cast<const bf16x8*>(b_rd + t * W3_BTAP + s * W3_BPLANE + (KK) * 32);            \
  }
  constexpr int SA[6] = {2, 1, 0, 1, 0, 0}, SB[6] = {0, 1, 2, 0, 1, 0};
#define STK_W3_MFMAS                                                                                       \
  _Pragma("unroll") for (int pr = 0; pr < 6; ++pr) _Pragma("unroll") for (int i = 0; i < 2; ++i)              \
    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                            \
      acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][SA[pr]], b[t][SB[pr]], acc[i][t], 0, 0, 0);

#pragma unroll
  for (int g = 0; g < 24; ++g) { al.ld(g, p, q, c_begin); bl.ld(g, p, c_begin); }
  {
    const int c1 = min(c_begin + 1, c_last);
#pragma unroll
    for (int g = 0; g < 24; ++g) { al.st(g, As); bl.st(g, Bs); }
#pragma unroll
    for (int g = 0; g < 24; ++g) { al.ld(g, p, q, c1); bl.ld(g, p, c1); }
  }
  bf16x8 a[2][3], b[3][3];
  for (int c = c_begin; c < c_last; ++c) {
    __syncthreads();                                   // chunk c is in LDS
    STK_W3_FRAGS(0)
    STK_W3_MFMAS
    STK_W3_FRAGS(1)
    __syncthreads();                                   // nobody reads LDS any more
    const int c2 = min(c + 2, c_last);
    STK_W3_MFMAS
#pragma unroll
    for (int g = 0; g < 24; ++g) al.st(g, As);
#pragma unroll
    for (int g = 0; g < 24; ++g) bl.st(g, Bs);
#pragma unroll
    for (int g = 0; g < 24; ++g) al.ld(g, p, q, c2);
#pragma unroll
    for (int g = 0; g < 24; ++g) bl.ld(g, p, c2);
#pragma unroll
    for (int g = 0; g < 36; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);     // VALU
      if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
    }
  }
  __syncthreads();
  STK_W3_FRAGS(0)
  STK_W3_MFMAS
  STK_W3_FRAGS(1)
  STK_W3_MFMAS
#undef STK_W3_MFMAS
#undef STK_W3_FRAGS

  // partial slab of split zs as [tap][Cout][Cin] (lanes = ci, contiguous); splitk_reduce_kernel re-lays it out
  float* slab = p.part + (long)zs * p.part_stride;
  const int n = n0 + wn0 + fc;
  if (n < p.Cin) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm0 + i * 32 + 4 * fk + igemm::strip_row(e);
          if (m < p.Cout) slab[((long)(kh * 3 + t) * p.Cout + m) * p.Cin + n] = acc[i][t][e];
        }
  }
}

inline int pad128(int v) { return (v + 127) / 128 * 128; }
// bytes of prepared weights for an M x Kc layer
inline long wp_bytes(int M, int Kc, int taps) { return 3L * taps * pad128(M) * Kc * 2; }

}  // namespace x3
