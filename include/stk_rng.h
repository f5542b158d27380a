/*
 * stk_rng.h -- the counter-based RNG shared by libstk (HIP) and the oracle's C restatement.
 *
 * u(seed, i) = 24 bits of a splitmix64-style finaliser of (seed, i >> 1), scaled to [0,1) (input pipeline draws).
 * Dropout keeps element i iff its 16-bit field of the mix of (seed, i >> 2) is >= round(65536 p) (stk_keep below).  Because both libraries use this exact integer
 * function, a dropout mask is a pure function of (seed, flat element index) and the HIP path and
 * the CPU checker agree bit-for-bit on which elements are dropped.
 * (The reference uses torch's Philox-based nn.Dropout, models/layerspp.py:245,278; CPU and GPU
 * streams of torch differ from each other as well, so parity on masks can only be statistical
 * against the reference and exact between our two libraries.)
 */
#ifndef STK_RNG_H
#define STK_RNG_H

#if defined(__HIPCC__)
#define STK_HD __host__ __device__ __forceinline__
#else
#define STK_HD static inline
#endif

STK_HD unsigned long long stk_mix64(unsigned long long seed, unsigned long long i) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (i + 1ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}

/* One 64-bit mix serves two consecutive indices (24 bits each): kernels that draw for runs of consecutive elements
 * pay the three 64-bit multiplies once per pair (the compiler merges the two identical stk_mix64 calls). */
STK_HD float stk_uniform(unsigned long long seed, unsigned long long i) {
  const unsigned long long z = stk_mix64(seed, i >> 1);
  return (float)((i & 1ULL) ? (z >> 8) & 0xFFFFFFULL : z >> 40) * (1.0f / 16777216.0f);
}

/* Dropout draws.  16 bits per element, FOUR consecutive elements (i >> 2) share one 64-bit mix: element i is kept iff
 * its 16-bit field is >= stk_drop_threshold(p), i.e. it is dropped with probability round(65536 p) / 65536 (p = 0.1:
 * 0.100006).  The three 64-bit multiplies of a mix are ~12 quarter-rate integer instructions on gfx950; with a 24-bit
 * draw per element (two per mix) they cost a GroupNorm + SiLU + Dropout kernel more than everything else it computes
 * (C128 @ 32x32, batch 128: 42.9 us with dropout, 29.7 us without). */
STK_HD unsigned stk_drop_threshold(float p) { return (unsigned)(p * 65536.0f + 0.5f); }
STK_HD unsigned stk_drop_field(unsigned long long z, unsigned lane) { return (unsigned)(z >> (16u * lane)) & 0xFFFFu; }
STK_HD int stk_keep(unsigned long long seed, unsigned long long i, unsigned thr) {
  return stk_drop_field(stk_mix64(seed, i >> 2), (unsigned)(i & 3ULL)) >= thr;
}

#endif /* STK_RNG_H */
