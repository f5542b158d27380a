// conv_x3.h -- what is left of the round-1 bf16 three-way-split ("x3") convolution path: the pieces the fp16 two-way-split
// kernels (conv_x2.h, conv_x2d.h, conv_x2w.h) share with it -- the activation-operand descriptor (Src), raw buffer loads, the
// fp32 row loaders of the weight gradients (RowsLoader, Rows3Loader) and the sizing helpers.  Included by conv.hip inside its
// anonymous namespace.  The x3 GEMM / weight-gradient kernels themselves (six bf16 MFMAs per fp32 product) were the fallback of a
// debugging switch since round 2 and were retired in round 6 (git history; DESIGN.md "Retired").
//
// K is ordered (32-channel group, tap, channel in group): a 32-wide k chunk is one tap and 32 consecutive channels,
// so the tap (hence the halo test and the pixel shift) is uniform over the chunk and a thread's 16 loads differ only
// by a scalar channel-plane offset; and the nine chunks of a channel group follow each other, so the nine shifted
// reads of the same 32 x (128 + halo) activation patch hit L1 / L2.
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

inline int pad128(int v) { return (v + 127) / 128 * 128; }
// bytes of prepared weights for an M x Kc layer
inline long wp_bytes(int M, int Kc, int taps) { return 3L * taps * pad128(M) * Kc * 2; }

}  // namespace x3
