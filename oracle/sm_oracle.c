/*
 * sm_oracle.c -- CPU ORACLE for the sparsify.me hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libsparsifyme.so) never links, loads or falls back to anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" for the 2:4 magnitude prune / compress / matmul steps.
 * The reference (owensgroup/sparsify.me) delegates those to closed vendor libraries that are
 * absent from /root/reference (cuSPARSELt 0.1.0 -- .MISSING_LARGE_BLOBS:1, header version macros
 * examples/libcusparse_lt/include/cusparseLt.h:24-30; cuBLAS / cuSPARSE from "CUDA >= 11",
 * README.md:25) and ships no tests, golden vectors or expected outputs.  What IS pinned:
 *   - sm_sparsify_positional_ref restates include/sparsify.me/sparsify.hxx:32-81 statement by
 *     statement (the only device code the reference contains) and is checked against
 *     hand-derived known answers in tests/golden/.
 *   - operand roles, layouts and leading dimensions follow the reference's call sites:
 *     include/sparsify.me/spmma.hxx:40-64,86,100-103,112-113 (row-major A m x k pruned in place,
 *     B k x n, C m x n) and include/sparsify.me/gemm.hxx:80-81 (column-major, lda=m ldb=k ldc=m).
 *   - the prune semantics follow the published vendor contract (STRIP: keep the two largest
 *     |x| of every 1x4 strip along k; TILE: in every 4x4 tile keep 8 elements, exactly two per
 *     row and per column, with maximum L1 norm), as documented offline in
 *     /opt/rocm/include/hipsparselt/hipsparselt.h:257-258.
 * Choices nothing upstream pins (tie-breaks, NaN order, metadata packing, padding) are frozen
 * here and in DESIGN.md; the HIP kernels must reproduce them bit for bit.
 *
 * Plain C99 + optional OpenMP; no dependencies.  Build: see oracle/Makefile.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SM_OK 0
#define SM_ERR_ARG 1

/* ------------------------------------------------------------------------------------------ */
/* fp16 <-> fp32 (IEEE binary16, round to nearest even), bit exact, no compiler support needed */
/* ------------------------------------------------------------------------------------------ */
static float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        ++e;
        man <<= 1;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static uint16_t f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((ax >> 13) & 0x3ffu));
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x33000001u) { /* < 2^-25 (or == 2^-25 ties to even 0) */
    return (uint16_t)sign;
  }
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t man = (ax & 0x7fffffu) | 0x800000u;
  if (e < -14) { /* subnormal half */
    int shift = 13 + (-14 - e); /* 14..24 */
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    return (uint16_t)(sign | q);
  }
  uint32_t q = ((uint32_t)(e + 15) << 10) | ((man >> 13) & 0x3ffu);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ++q; /* carry may bump exponent: fine */
  return (uint16_t)(sign | q);
}

/* bfloat16 (the upper half of an IEEE binary32) <-> fp32, round to nearest even; a NaN keeps its sign and upper
 * payload bits and is made quiet, infinities and the exponent range carry over unchanged. */
static float bf2f(uint16_t h) {
  const uint32_t bits = (uint32_t)h << 16;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}
static uint16_t f2bf(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x0040u);
  x += 0x7fffu + ((x >> 16) & 1u); /* ties to even; a carry into the exponent (up to inf) is the right result */
  return (uint16_t)(x >> 16);
}

/* ------------------------------------------------------------------------------------------ */
/* (a1) positional sparsify -- restates include/sparsify.me/sparsify.hxx:32-81                 */
/* ------------------------------------------------------------------------------------------ */
/*
 * weights: m*n elements of elt_bytes each (2, 4 or 8); only zeros are ever stored, so the
 * element type matters through its size alone.  mask: m*n values of size_t (uint64_t here).
 * Follows the reference exactly:
 *   tile_m = m / blk_m, tile_n = n / blk_n               (sparsify.hxx:38-39, integer division)
 *   number_of_zeros_per_block = floor(blk_size * sf)     (sparsify.hxx:42, float arithmetic)
 *   mask[0 .. m*n) = 1                                   (sparsify.hxx:71)
 *   for blk_idx in [0, tile_m*tile_n): global_idx = blk_idx * blk_size; visit h-major over
 *   (h < blk_m, w < blk_n), idx = global_idx + h + w*blk_n, zero weights[idx] and mask[idx]
 *   until number_of_zeros_per_block have been written (sparsify.hxx:43-68).
 * Note the inner `break` (sparsify.hxx:55-56) leaves only the w loop; the h loop then re-tests
 * the same condition at once, so the effect is "stop after nz stores".
 * For block shapes other than 2x2 the reference's idx leaves its block (sparsify.hxx:60); this
 * restatement reproduces that arithmetic but refuses (SM_ERR_ARG) a call whose largest idx
 * would fall outside the m*n buffer, where the reference would write out of bounds.
 */
int sm_sparsify_positional_ref(void* weights, uint64_t* mask, size_t m, size_t n, size_t elt_bytes,
                               size_t blk_m, size_t blk_n, float sparsity_factor) {
  if (!weights || !mask || blk_m == 0 || blk_n == 0) return SM_ERR_ARG;
  if (elt_bytes != 2 && elt_bytes != 4 && elt_bytes != 8) return SM_ERR_ARG;
  const size_t blk_size = blk_m * blk_n;
  const size_t tile_m = m / blk_m, tile_n = n / blk_n;
  const size_t nblk = tile_m * tile_n;
  const float nzf = floorf((float)blk_size * sparsity_factor);
  size_t nz = nzf <= 0.0f ? 0 : (size_t)nzf;
  if (nz > blk_size) nz = blk_size;
  /* largest index any block can touch */
  if (nblk > 0 && nz > 0) {
    size_t max_off = 0, cnt = 0;
    for (size_t h = 0; h < blk_m && cnt < nz; ++h)
      for (size_t w = 0; w < blk_n && cnt < nz; ++w, ++cnt) {
        size_t off = h + w * blk_n;
        if (off > max_off) max_off = off;
      }
    if ((nblk - 1) * blk_size + max_off >= m * n) return SM_ERR_ARG;
  }
  for (size_t i = 0; i < m * n; ++i) mask[i] = 1;
  unsigned char* wb = (unsigned char*)weights;
  for (size_t blk = 0; blk < nblk; ++blk) {
    const size_t g = blk * blk_size;
    size_t sparsified = 0;
    for (size_t h = 0; h < blk_m; ++h) {
      for (size_t w = 0; w < blk_n; ++w) {
        if (sparsified == nz) break;
        const size_t idx = g + h + w * blk_n;
        memset(wb + idx * elt_bytes, 0, elt_bytes);
        mask[idx] = 0;
        ++sparsified;
      }
    }
  }
  return SM_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* 2:4 selection rules (frozen here; see DESIGN.md "2:4 format")                               */
/* ------------------------------------------------------------------------------------------ */
/*
 * Magnitude key of an element = its bit pattern with the sign bit cleared, compared as an
 * unsigned integer.  For finite values this is exactly |x| ordering; +0 == -0 (key 0);
 * inf > every finite value; NaN keys exceed inf (a NaN is always kept -- it is never silently
 * dropped).  Ties keep the lower k index.
 */
static inline uint32_t key16(uint16_t v) { return v & 0x7fffu; }
static inline uint32_t key32(uint32_t v) { return v & 0x7fffffffu; }
/* int8 (extension): |x| of a signed byte, 0 .. 128 */
static inline uint32_t key8(uint8_t v) { const int x = (int)(int8_t)v; return (uint32_t)(x < 0 ? -x : x); }

/* STRIP rule on 4 keys: returns nibble p0 | p1 << 2 with p0 < p1 the kept positions. */
static inline unsigned strip_select(const uint32_t key[4]) {
  unsigned keep[2], nk = 0;
  for (unsigned i = 0; i < 4; ++i) {
    unsigned rank = 0; /* number of elements that beat element i */
    for (unsigned j = 0; j < 4; ++j) {
      if (j == i) continue;
      if (key[j] > key[i] || (key[j] == key[i] && j < i)) ++rank;
    }
    if (rank < 2) keep[nk++] = i;
  }
  return keep[0] | (keep[1] << 2);
}

/* The 6 column pairs of a row of 4, in the frozen enumeration order. */
static const unsigned char PAIR_C0[6] = {0, 0, 0, 1, 1, 2};
static const unsigned char PAIR_C1[6] = {1, 2, 3, 2, 3, 3};

/*
 * TILE rule on a 4x4 tile of magnitudes (float; NaN replaced by +inf by the caller).
 * Row r keeps column pair pr[r] in 0..5; a pattern (p0,p1,p2,p3) is valid when every column is kept exactly twice
 * (90 of the 1296).  With s_r[p] = mag[r][c0] + mag[r][c1] (fp32), the frozen rule is stated in two levels:
 *   for every choice (p0,p1) of rows 0-1, in lexicographic order:
 *     completion(p0,p1) = the valid (p2,p3) with the greatest fp32 s2[p2] + s3[p3], the first in lexicographic
 *                         order among equals;
 *     score(p0,p1)      = (s0[p0] + s1[p1]) + (s2[p2] + s3[p3]) of that completion, in fp32;
 *   the first (p0,p1) with the strictly greatest score wins, with its completion.
 * fp32 addition is monotone, so the winning score is the maximum of (s0+s1)+(s2+s3) over all 90 patterns, and in
 * exact arithmetic (integer data, int8) the winner is the lexicographically first maximal pattern -- the definition
 * of round 1 (tile_select_exhaustive below, kept to check exactly that).  The two can differ only when fp32
 * rounding makes patterns with different (s2+s3) collide on the same total.  Stated this way the rule needs 36 + 36
 * pair-of-row sums, 19 completion maxima (choices that use the columns equally share their completions) and 36
 * scores instead of 90 four-term totals: that is what the kernels compute (csrc/tile_rule.inc).
 */
static int tile_valid(unsigned p0, unsigned p1, unsigned p2, unsigned p3) {
  unsigned cnt[4] = {0, 0, 0, 0};
  ++cnt[PAIR_C0[p0]]; ++cnt[PAIR_C1[p0]];
  ++cnt[PAIR_C0[p1]]; ++cnt[PAIR_C1[p1]];
  ++cnt[PAIR_C0[p2]]; ++cnt[PAIR_C1[p2]];
  ++cnt[PAIR_C0[p3]]; ++cnt[PAIR_C1[p3]];
  return cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2;
}

static void tile_select(const float mag[4][4], unsigned pr[4]) {
  float s[4][6];
  for (int r = 0; r < 4; ++r)
    for (int p = 0; p < 6; ++p) s[r][p] = mag[r][PAIR_C0[p]] + mag[r][PAIR_C1[p]];
  float best = -1.0f;
  pr[0] = pr[1] = pr[2] = pr[3] = 0;
  for (unsigned p0 = 0; p0 < 6; ++p0)
    for (unsigned p1 = 0; p1 < 6; ++p1) {
      float bb = -1.0f;
      unsigned b2 = 0, b3 = 0;
      for (unsigned p2 = 0; p2 < 6; ++p2)
        for (unsigned p3 = 0; p3 < 6; ++p3) {
          if (!tile_valid(p0, p1, p2, p3)) continue;
          const float b = s[2][p2] + s[3][p3];
          if (b > bb) {
            bb = b;
            b2 = p2; b3 = p3;
          }
        }
      const float sc = (s[0][p0] + s[1][p1]) + bb;
      if (sc > best) {
        best = sc;
        pr[0] = p0; pr[1] = p1; pr[2] = b2; pr[3] = b3;
      }
    }
}

/* Round 1's statement of the rule: first strictly greatest of all 90 totals in lexicographic order.  Test aid. */
static void tile_select_exhaustive(const float mag[4][4], unsigned pr[4]) {
  float s[4][6];
  for (int r = 0; r < 4; ++r)
    for (int p = 0; p < 6; ++p) s[r][p] = mag[r][PAIR_C0[p]] + mag[r][PAIR_C1[p]];
  float best = -1.0f;
  pr[0] = pr[1] = pr[2] = pr[3] = 0;
  for (unsigned p0 = 0; p0 < 6; ++p0)
    for (unsigned p1 = 0; p1 < 6; ++p1)
      for (unsigned p2 = 0; p2 < 6; ++p2)
        for (unsigned p3 = 0; p3 < 6; ++p3) {
          if (!tile_valid(p0, p1, p2, p3)) continue;
          const float sc = (s[0][p0] + s[1][p1]) + (s[2][p2] + s[3][p3]);
          if (sc > best) {
            best = sc;
            pr[0] = p0; pr[1] = p1; pr[2] = p2; pr[3] = p3;
          }
        }
}

/* Both statements on one tile of fp32 magnitudes: keep masks (bit 4*row + col) and scores.  Test aid. */
int sm_tile_select_both_ref(const float* mag16, unsigned* mask_two_level, unsigned* mask_exhaustive, float* score_two_level,
                            float* score_exhaustive) {
  float mag[4][4];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) mag[r][c] = mag16[4 * r + c];
  unsigned pa[4], pb[4];
  tile_select(mag, pa);
  tile_select_exhaustive(mag, pb);
  unsigned ma = 0, mb = 0;
  float sa = 0.0f, sb = 0.0f, ra[4], rb[4];
  for (int r = 0; r < 4; ++r) {
    ma |= (1u << (4 * r + PAIR_C0[pa[r]])) | (1u << (4 * r + PAIR_C1[pa[r]]));
    mb |= (1u << (4 * r + PAIR_C0[pb[r]])) | (1u << (4 * r + PAIR_C1[pb[r]]));
    ra[r] = mag[r][PAIR_C0[pa[r]]] + mag[r][PAIR_C1[pa[r]]];
    rb[r] = mag[r][PAIR_C0[pb[r]]] + mag[r][PAIR_C1[pb[r]]];
  }
  sa = (ra[0] + ra[1]) + (ra[2] + ra[3]);
  sb = (rb[0] + rb[1]) + (rb[2] + rb[3]);
  *mask_two_level = ma; *mask_exhaustive = mb; *score_two_level = sa; *score_exhaustive = sb;
  return SM_OK;
}

static inline float mag16(uint16_t v) {
  uint32_t k = key16(v);
  return k > 0x7c00u ? INFINITY : h2f((uint16_t)k);
}
static inline float magbf(uint16_t v) {
  uint32_t k = key16(v);
  return k > 0x7f80u ? INFINITY : bf2f((uint16_t)k);
}
static inline float mag32(uint32_t v) {
  uint32_t k = key32(v);
  if (k > 0x7f800000u) return INFINITY;
  float f;
  memcpy(&f, &k, 4);
  return f;
}

/* ------------------------------------------------------------------------------------------ */
/* (a2) prune to 2:4 -- the step include/sparsify.me/spmma.hxx:86-87 delegates to the vendor    */
/* ------------------------------------------------------------------------------------------ */
/*
 * A_in, A_out: row-major m x k, leading dimension ld (elements), may alias (the reference
 * prunes in place: spmma.hxx:86 passes dA as input and output).  alg: 0 = TILE, 1 = STRIP
 * (numbering of cusparseLtPruneAlg_t as used at spmma.hxx:86).  Elements outside the kept set
 * are written as +0; kept elements are copied bit for bit.  A ragged last strip (k % 4 != 0)
 * or tile (m % 4 != 0) is completed with virtual zeros placed AFTER the real elements.
 */
#define DEFINE_PRUNE(NAME, T, KEY, MAG)                                                          \
  int NAME(const T* A_in, T* A_out, size_t m, size_t k, size_t ld, int alg) {                    \
    if (!A_in || !A_out || ld < k || (alg != 0 && alg != 1)) return SM_ERR_ARG;                  \
    if (alg == 1) {                                                                              \
      for (size_t i = 0; i < m; ++i)                                                             \
        for (size_t c = 0; c < k; c += 4) {                                                      \
          uint32_t key[4];                                                                       \
          T v[4];                                                                                \
          for (unsigned t = 0; t < 4; ++t) {                                                     \
            v[t] = (c + t < k) ? A_in[i * ld + c + t] : (T)0;                                    \
            key[t] = KEY(v[t]);                                                                  \
          }                                                                                      \
          const unsigned nib = strip_select(key);                                                \
          const unsigned p0 = nib & 3u, p1 = nib >> 2;                                           \
          for (unsigned t = 0; t < 4 && c + t < k; ++t)                                          \
            A_out[i * ld + c + t] = (t == p0 || t == p1) ? v[t] : (T)0;                          \
        }                                                                                        \
      return SM_OK;                                                                              \
    }                                                                                            \
    for (size_t i0 = 0; i0 < m; i0 += 4)                                                         \
      for (size_t c = 0; c < k; c += 4) {                                                        \
        float mag[4][4];                                                                         \
        T v[4][4];                                                                               \
        for (unsigned r = 0; r < 4; ++r)                                                         \
          for (unsigned t = 0; t < 4; ++t) {                                                     \
            v[r][t] = (i0 + r < m && c + t < k) ? A_in[(i0 + r) * ld + c + t] : (T)0;            \
            mag[r][t] = MAG(v[r][t]);                                                            \
          }                                                                                      \
        unsigned pr[4];                                                                          \
        tile_select(mag, pr);                                                                    \
        for (unsigned r = 0; r < 4 && i0 + r < m; ++r)                                           \
          for (unsigned t = 0; t < 4 && c + t < k; ++t)                                          \
            A_out[(i0 + r) * ld + c + t] =                                                       \
                (t == PAIR_C0[pr[r]] || t == PAIR_C1[pr[r]]) ? v[r][t] : (T)0;                   \
      }                                                                                          \
    return SM_OK;                                                                                \
  }

DEFINE_PRUNE(sm_prune24_f16_ref, uint16_t, key16, mag16)
DEFINE_PRUNE(sm_prune24_f32_bits_ref, uint32_t, key32, mag32)
/* bfloat16 (extension of the build): magnitudes order by bit pattern exactly as fp16's do, so
 * the STRIP rule is the same function of the bits; the TILE rule sums bfloat16 magnitudes */
DEFINE_PRUNE(sm_prune24_bf16_ref, uint16_t, key16, magbf)
static inline float mag8(uint8_t v) { return (float)key8(v); }
DEFINE_PRUNE(sm_prune24_i8_ref, uint8_t, key8, mag8)

int sm_prune24_f32_ref(const float* A_in, float* A_out, size_t m, size_t k, size_t ld, int alg) {
  return sm_prune24_f32_bits_ref((const uint32_t*)A_in, (uint32_t*)A_out, m, k, ld, alg);
}

/* ------------------------------------------------------------------------------------------ */
/* (a2) prune check -- the step spmma.hxx:88 delegates to cusparseLtSpMMAPruneCheck             */
/* ------------------------------------------------------------------------------------------ */
/* *valid = 0 iff every 1x4 strip along k holds at most two values that compare != 0
 * (so -0 counts as zero and NaN as non-zero); otherwise 1.  spmma.hxx:92-94 treats != 0 as
 * "Incorrect pruning results." */
#define DEFINE_CHECK(NAME, T, ISNZ)                                                    \
  int NAME(const T* A, size_t m, size_t k, size_t ld, int* valid) {                    \
    if (!A || !valid || ld < k) return SM_ERR_ARG;                                     \
    int bad = 0;                                                                       \
    for (size_t i = 0; i < m; ++i)                                                     \
      for (size_t c = 0; c < k; c += 4) {                                              \
        unsigned nnz = 0;                                                              \
        for (unsigned t = 0; t < 4 && c + t < k; ++t) nnz += ISNZ(A[i * ld + c + t]);  \
        if (nnz > 2) bad = 1;                                                          \
      }                                                                                \
    *valid = bad;                                                                      \
    return SM_OK;                                                                      \
  }
#define ISNZ16(v) (((v) & 0x7fffu) != 0)
#define ISNZ32(v) (((v) & 0x7fffffffu) != 0)
DEFINE_CHECK(sm_prune24_check_f16_ref, uint16_t, ISNZ16)
DEFINE_CHECK(sm_prune24_check_f32_bits_ref, uint32_t, ISNZ32)
#define ISNZ8(v) ((v) != 0)
DEFINE_CHECK(sm_prune24_check_i8_ref, uint8_t, ISNZ8)
int sm_prune24_check_f32_ref(const float* A, size_t m, size_t k, size_t ld, int* valid) {
  return sm_prune24_check_f32_bits_ref((const uint32_t*)A, m, k, ld, valid);
}

/* ------------------------------------------------------------------------------------------ */
/* (a3) compress -- the step spmma.hxx:100-103 delegates to cusparseLtSpMMACompress             */
/* ------------------------------------------------------------------------------------------ */
/*
 * Compressed blob of a batch of `batch` row-major m x k matrices (M = batch*m rows in all):
 *   kc          = k rounded up to a multiple of 64
 *   values      : [kc/64][M][32] elements at byte 0 (STAGE-major like the metadata, see val_index);
 *                 row R = b*m + i; the two kept elements of the strip covering columns 4q..4q+3
 *                 sit at [q/16][R][2(q%16)], [q/16][R][2(q%16)+1] in k order; strips at or
 *                 beyond k hold +0
 *   metadata    : [kc/64][M][8] bytes (stage-major, see meta_index) at byte
 *                 meta_off = round_up(M*(kc/2)*elt, 256); byte [q/16][R][(q%16)/2] holds strip q's
 *                 nibble in bits 4*(q&1)..4*(q&1)+3;
 *                 nibble = p0 | p1 << 2, p0 < p1 the kept positions (the 2-bit codes
 *                 v_smfmac_* consumes, see profiles/probe_gfx950_r01.txt); padding strips
 *                 carry 0x4 (positions 0,1)
 *   total bytes = meta_off + round_up(M*(kc/8), 256)
 * The kept positions are chosen by the STRIP rule applied to the input strip, so for an input
 * already pruned to 2:4 they are its non-zeros (completed with the lowest free positions when a
 * strip has fewer than two), and compress(A) == compress(prune_strip(A)) for any A.
 */
static size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

/* Byte of the metadata section that holds strip q (dense columns 4q..4q+3) of blob row R, M rows in all.
 * The section is STAGE-major: plane s = q / 16 covers dense k 64s..64s+63 of every row, 8 bytes per row:
 * [kc/64][M][8].  (A 128-row x 64-k tile of the matmul is then 1 KiB of contiguous metadata -- one
 * 8-cache-line DMA -- instead of 128 eight-byte pieces on 128 lines: profiles/stamp_r01.txt.) */
static size_t meta_index(size_t M, size_t R, size_t q) { return ((q / 16) * M + R) * 8 + (q % 16) / 2; }

/* Element index (values section) of the FIRST kept element of strip q of blob row R; the second follows it.
 * Also stage-major: plane s = q / 16 holds the 32 kept elements of dense k 64s..64s+63 of every row,
 * [kc/64][M][32].  (The 128-row x 64-k A tile of the fp16 matmul is then 8 KiB of contiguous values --
 * 64 whole 128-byte lines -- instead of 128 half lines at a row pitch: the matmul is bound by the
 * number of line requests its CU can issue, profiles/stamp_r01.txt.) */
static size_t val_index(size_t M, size_t R, size_t q) { return ((q / 16) * M + R) * 32 + 2 * (q % 16); }

int sm_compress24_layout(size_t m, size_t k, size_t elt_bytes, size_t batch, size_t* kc_out,
                         size_t* meta_off_out, size_t* total_out) {
  if (elt_bytes != 1 && elt_bytes != 2 && elt_bytes != 4) return SM_ERR_ARG;
  const size_t kc = round_up(k, 64), M = m * batch;
  const size_t meta_off = round_up(M * (kc / 2) * elt_bytes, 256);
  if (kc_out) *kc_out = kc;
  if (meta_off_out) *meta_off_out = meta_off;
  if (total_out) *total_out = meta_off + round_up(M * (kc / 8), 256);
  return SM_OK;
}

int sm_compress24_size_ref(size_t m, size_t k, size_t elt_bytes, size_t batch, size_t* bytes) {
  if (!bytes) return SM_ERR_ARG;
  return sm_compress24_layout(m, k, elt_bytes, batch, NULL, NULL, bytes);
}

#define DEFINE_COMPRESS(NAME, T, KEY)                                                            \
  int NAME(const T* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob) { \
    if (!A || !blob || ld < k) return SM_ERR_ARG;                                                \
    size_t kc, meta_off, total;                                                                  \
    sm_compress24_layout(m, k, sizeof(T), batch, &kc, &meta_off, &total);                        \
    memset(blob, 0, total);                                                                      \
    T* vals = (T*)blob;                                                                          \
    unsigned char* meta = (unsigned char*)blob + meta_off;                                       \
    const size_t M = m * batch;                                                                  \
    for (size_t b = 0; b < batch; ++b)                                                           \
      for (size_t i = 0; i < m; ++i) {                                                           \
        const size_t R = b * m + i;                                                              \
        const T* row = A + b * strideA + i * ld;                                                 \
        for (size_t q = 0; q < kc / 4; ++q) {                                                    \
          unsigned nib = 0x4u;                                                                   \
          T v[4] = {0, 0, 0, 0};                                                                 \
          if (4 * q < k) {                                                                       \
            uint32_t key[4];                                                                     \
            for (unsigned t = 0; t < 4; ++t) {                                                   \
              v[t] = (4 * q + t < k) ? row[4 * q + t] : (T)0;                                    \
              key[t] = KEY(v[t]);                                                                \
            }                                                                                    \
            nib = strip_select(key);                                                             \
          }                                                                                      \
          vals[val_index(M, R, q)] = v[nib & 3u];                                                \
          vals[val_index(M, R, q) + 1] = v[nib >> 2];                                            \
          meta[meta_index(M, R, q)] |= (unsigned char)(nib << (4 * (q & 1)));                    \
        }                                                                                        \
      }                                                                                          \
    return SM_OK;                                                                                \
  }
DEFINE_COMPRESS(sm_compress24_f16_ref, uint16_t, key16)
DEFINE_COMPRESS(sm_compress24_f32_bits_ref, uint32_t, key32)
DEFINE_COMPRESS(sm_compress24_i8_ref, uint8_t, key8)
int sm_compress24_f32_ref(const float* A, size_t m, size_t k, size_t ld, size_t batch,
                          size_t strideA, void* blob) {
  return sm_compress24_f32_bits_ref((const uint32_t*)A, m, k, ld, batch, strideA, blob);
}

/* Inverse of compress: dense row-major [batch][m][ld] with +0 at dropped positions. */
#define DEFINE_DECOMPRESS(NAME, T)                                                               \
  int NAME(const void* blob, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, T* A) { \
    if (!A || !blob || ld < k) return SM_ERR_ARG;                                                \
    size_t kc, meta_off, total;                                                                  \
    sm_compress24_layout(m, k, sizeof(T), batch, &kc, &meta_off, &total);                        \
    const T* vals = (const T*)blob;                                                              \
    const unsigned char* meta = (const unsigned char*)blob + meta_off;                           \
    for (size_t b = 0; b < batch; ++b)                                                           \
      for (size_t i = 0; i < m; ++i) {                                                           \
        const size_t R = b * m + i;                                                              \
        T* row = A + b * strideA + i * ld;                                                       \
        for (size_t c = 0; c < k; ++c) row[c] = 0;                                               \
        for (size_t q = 0; 4 * q < k; ++q) {                                                     \
          const unsigned nib = (meta[meta_index(m * batch, R, q)] >> (4 * (q & 1))) & 0xfu;      \
          const unsigned p0 = nib & 3u, p1 = nib >> 2;                                           \
          if (4 * q + p0 < k) row[4 * q + p0] = vals[val_index(m * batch, R, q)];                \
          if (4 * q + p1 < k) row[4 * q + p1] = vals[val_index(m * batch, R, q) + 1];            \
        }                                                                                        \
      }                                                                                          \
    return SM_OK;                                                                                \
  }
DEFINE_DECOMPRESS(sm_decompress24_f16_ref, uint16_t)
DEFINE_DECOMPRESS(sm_decompress24_f32_ref, float)
DEFINE_DECOMPRESS(sm_decompress24_i8_ref, uint8_t)

/* ------------------------------------------------------------------------------------------ */
/* (a4) 2:4 sparse x dense matmul -- the step spmma.hxx:112-113 delegates to cusparseLtMatmul   */
/* ------------------------------------------------------------------------------------------ */
/*
 * C_b (m x n, row-major, ld n) = alpha * A_b * B_b + beta * C_b for b < batch, A_b the b-th
 * matrix of the compressed blob, B_b = B + b*strideB (k x n row-major, ld n; strideB = 0 shares
 * one B), C_b = C + b*strideC.  Accumulation in fp64 in k order; one rounding to the output type.
 * (spmma.hxx:40-64: row-major, ld(A)=k, ld(B)=ld(C)=n, opA=opB=N.)
 */
static double ld16(const void* p, size_t i) { return (double)h2f(((const uint16_t*)p)[i]); }
static double ld32(const void* p, size_t i) { return (double)((const float*)p)[i]; }
static void st16(void* p, size_t i, double v) { ((uint16_t*)p)[i] = f2h((float)v); }
static void st32(void* p, size_t i, double v) { ((float*)p)[i] = (float)v; }

static int spmma_ref_impl(const void* blob, const void* B, void* C, size_t m, size_t n, size_t k,
                          size_t batch, size_t strideB, size_t strideC, float alpha, float beta,
                          size_t elt, double (*ld)(const void*, size_t),
                          void (*st)(void*, size_t, double)) {
  if (!blob || !B || !C) return SM_ERR_ARG;
  size_t kc, meta_off, total;
  sm_compress24_layout(m, k, elt, batch, &kc, &meta_off, &total);
  const unsigned char* meta = (const unsigned char*)blob + meta_off;
  double* acc = (double*)malloc(sizeof(double) * (n ? n : 1));
  if (!acc) return SM_ERR_ARG;
  for (size_t b = 0; b < batch; ++b)
    for (size_t i = 0; i < m; ++i) {
      const size_t R = b * m + i;
      for (size_t j = 0; j < n; ++j) acc[j] = 0.0;
      for (size_t q = 0; 4 * q < k; ++q) {
        const unsigned nib = (meta[meta_index(m * batch, R, q)] >> (4 * (q & 1))) & 0xfu;
        const unsigned pos[2] = {nib & 3u, nib >> 2};
        for (int t = 0; t < 2; ++t) {
          const size_t kk = 4 * q + pos[t];
          if (kk >= k) continue;
          const double a = ld(blob, val_index(m * batch, R, q) + (size_t)t);
          if (a == 0.0) continue; /* exact: a zero contributes nothing for finite B */
          const size_t brow = b * strideB + kk * n;
          for (size_t j = 0; j < n; ++j) acc[j] += a * ld(B, brow + j);
        }
      }
      for (size_t j = 0; j < n; ++j) {
        const size_t ci = b * strideC + i * n + j;
        double v = (double)alpha * acc[j];
        if (beta != 0.0f) v += (double)beta * ld(C, ci);
        st(C, ci, v);
      }
    }
  free(acc);
  return SM_OK;
}

int sm_spmma_f16_ref(const void* blob, const uint16_t* B, uint16_t* C, size_t m, size_t n, size_t k,
                     size_t batch, size_t strideB, size_t strideC, float alpha, float beta) {
  return spmma_ref_impl(blob, B, C, m, n, k, batch, strideB, strideC, alpha, beta, 2, ld16, st16);
}
static double ldbf(const void* p, size_t i) { return (double)bf2f(((const uint16_t*)p)[i]); }
static void stbf(void* p, size_t i, double v) { ((uint16_t*)p)[i] = f2bf((float)v); }
int sm_spmma_bf16_ref(const void* blob, const uint16_t* B, uint16_t* C, size_t m, size_t n, size_t k,
                      size_t batch, size_t strideB, size_t strideC, float alpha, float beta) {
  return spmma_ref_impl(blob, B, C, m, n, k, batch, strideB, strideC, alpha, beta, 2, ldbf, stbf);
}
/* int8 (extension): C_b (m x n int32, row-major) = A_b . B_b (+ C_b), exact; B_b is [n][k], K-CONTIGUOUS per output
 * column (the layout the int8 matrix instruction is fed in), B_b = B + b * strideB. */
int sm_spmma_i8_ref(const void* blob, const int8_t* B, int32_t* C, size_t m, size_t n, size_t k, size_t batch,
                    size_t strideB, size_t strideC, int accumulate) {
  if (!blob || !B || !C) return SM_ERR_ARG;
  size_t kc, meta_off, total;
  sm_compress24_layout(m, k, 1, batch, &kc, &meta_off, &total);
  const int8_t* vals = (const int8_t*)blob;
  const unsigned char* meta = (const unsigned char*)blob + meta_off;
  for (size_t b = 0; b < batch; ++b)
    for (size_t i = 0; i < m; ++i) {
      const size_t R = b * m + i;
      for (size_t j = 0; j < n; ++j) {
        int64_t acc = 0;
        const int8_t* bc = B + b * strideB + j * k;
        for (size_t q = 0; 4 * q < k; ++q) {
          const unsigned nib = (meta[meta_index(m * batch, R, q)] >> (4 * (q & 1))) & 0xfu;
          const unsigned pos[2] = {nib & 3u, nib >> 2};
          for (int t = 0; t < 2; ++t) {
            const size_t kk = 4 * q + pos[t];
            if (kk < k) acc += (int64_t)vals[val_index(m * batch, R, q) + (size_t)t] * (int64_t)bc[kk];
          }
        }
        const size_t ci = b * strideC + i * n + j;
        C[ci] = (int32_t)((accumulate ? (int64_t)C[ci] : 0) + acc);
      }
    }
  return SM_OK;
}
/* requantisation of the int32 product: saturate(round-to-nearest-even((float)acc * scale)) to a signed byte */
int sm_requant_i8_ref(const int32_t* acc, int8_t* out, size_t count, float scale) {
  if (!acc || !out) return SM_ERR_ARG;
  for (size_t i = 0; i < count; ++i) {
    float f = rintf(scale * (float)acc[i]);
    f = f < -128.0f ? -128.0f : (f > 127.0f ? 127.0f : f);
    out[i] = (int8_t)(int)f;
  }
  return SM_OK;
}
int sm_spmma_f32_ref(const void* blob, const float* B, float* C, size_t m, size_t n, size_t k,
                     size_t batch, size_t strideB, size_t strideC, float alpha, float beta) {
  return spmma_ref_impl(blob, B, C, m, n, k, batch, strideB, strideC, alpha, beta, 4, ld32, st32);
}

/* ------------------------------------------------------------------------------------------ */
/* (a5) dense batched GEMM -- the call include/sparsify.me/gemm.hxx:80-81 / 133-134 / 186-187   */
/* ------------------------------------------------------------------------------------------ */
/*
 * Pointer-array batch, COLUMN-major with the reference's fixed leading dimensions:
 * op(A_i) is m x k, op(B_i) k x n, C_i m x n, lda = m, ldb = k, ldc = m exactly as
 * gemm.hxx:80-81 passes them (so with a transpose flag set the stored matrix is read through
 * those same leading dimensions, as cuBLAS would).  ta/tb: 0 = N, 1 = T.
 * C_i = alpha * op(A_i) * op(B_i) + beta * C_i, fp64 accumulation, one rounding.
 */
static int gemm_ref_impl(void* const* A, void* const* B, void* const* C, size_t m, size_t n, size_t k,
                         size_t batch, int ta, int tb, double alpha, double beta,
                         double (*ld)(const void*, size_t), void (*st)(void*, size_t, double)) {
  if (!A || !B || !C) return SM_ERR_ARG;
  const size_t lda = m, ldb = k, ldc = m;
  for (size_t b = 0; b < batch; ++b) {
    const void* a = A[b];
    const void* bb = B[b];
    void* c = C[b];
    for (size_t j = 0; j < n; ++j)
      for (size_t i = 0; i < m; ++i) {
        double acc = 0.0;
        for (size_t l = 0; l < k; ++l) {
          const double av = ta ? ld(a, i * lda + l) : ld(a, l * lda + i);
          const double bv = tb ? ld(bb, l * ldb + j) : ld(bb, j * ldb + l);
          acc += av * bv;
        }
        double v = alpha * acc;
        if (beta != 0.0) v += beta * ld(c, j * ldc + i);
        st(c, j * ldc + i, v);
      }
  }
  return SM_OK;
}
static double ld64(const void* p, size_t i) { return ((const double*)p)[i]; }
static void st64(void* p, size_t i, double v) { ((double*)p)[i] = v; }

int sm_gemm_batched_f16_ref(void* const* A, void* const* B, void* const* C, size_t m, size_t n,
                            size_t k, size_t batch, int ta, int tb, float alpha, float beta) {
  return gemm_ref_impl(A, B, C, m, n, k, batch, ta, tb, alpha, beta, ld16, st16);
}
int sm_gemm_batched_f32_ref(void* const* A, void* const* B, void* const* C, size_t m, size_t n,
                            size_t k, size_t batch, int ta, int tb, float alpha, float beta) {
  return gemm_ref_impl(A, B, C, m, n, k, batch, ta, tb, alpha, beta, ld32, st32);
}
int sm_gemm_batched_f64_ref(void* const* A, void* const* B, void* const* C, size_t m, size_t n,
                            size_t k, size_t batch, int ta, int tb, double alpha, double beta) {
  return gemm_ref_impl(A, B, C, m, n, k, batch, ta, tb, alpha, beta, ld64, st64);
}

/* Row-major strided-batch dense GEMM (the layout spmma.hxx uses), fp64 accumulate: the dense
 * counterpart of sm_spmma_*_ref, used to check spmma(A,B) == gemm(prune(A),B). */
static int gemm_rm_impl(const void* A, const void* B, void* C, size_t m, size_t n, size_t k,
                        size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                        float alpha, float beta, double (*ld)(const void*, size_t),
                        void (*st)(void*, size_t, double)) {
  if (!A || !B || !C) return SM_ERR_ARG;
  double* acc = (double*)malloc(sizeof(double) * (n ? n : 1));
  if (!acc) return SM_ERR_ARG;
  for (size_t b = 0; b < batch; ++b)
    for (size_t i = 0; i < m; ++i) {
      for (size_t j = 0; j < n; ++j) acc[j] = 0.0;
      for (size_t l = 0; l < k; ++l) {
        const double a = ld(A, b * strideA + i * lda + l);
        if (a == 0.0) continue;
        for (size_t j = 0; j < n; ++j) acc[j] += a * ld(B, b * strideB + l * n + j);
      }
      for (size_t j = 0; j < n; ++j) {
        const size_t ci = b * strideC + i * n + j;
        double v = (double)alpha * acc[j];
        if (beta != 0.0f) v += (double)beta * ld(C, ci);
        st(C, ci, v);
      }
    }
  free(acc);
  return SM_OK;
}
int sm_gemm_rowmajor_f16_ref(const uint16_t* A, const uint16_t* B, uint16_t* C, size_t m, size_t n,
                             size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB,
                             size_t strideC, float alpha, float beta) {
  return gemm_rm_impl(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, ld16, st16);
}
int sm_gemm_rowmajor_bf16_ref(const uint16_t* A, const uint16_t* B, uint16_t* C, size_t m, size_t n,
                              size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB,
                              size_t strideC, float alpha, float beta) {
  return gemm_rm_impl(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, ldbf, stbf);
}
int sm_gemm_rowmajor_f32_ref(const float* A, const float* B, float* C, size_t m, size_t n, size_t k,
                             size_t lda, size_t batch, size_t strideA, size_t strideB,
                             size_t strideC, float alpha, float beta) {
  return gemm_rm_impl(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, ld32, st32);
}

/* ------------------------------------------------------------------------------------------ */
/* (a6) unstructured SpMM restatements: Blocked-ELL (spmm.hxx:57-67,107-110) and COO            */
/* (spmm.hxx:164-187).  Column-major dense operands as the reference declares them.            */
/* ------------------------------------------------------------------------------------------ */
/*
 * Blocked-ELL A (containers/ell.hxx:23-33): rows x cols, square blocks of block_size,
 * ell_cols stored columns per row; column_indices [blocked_rows][blocked_cols] (block-column id,
 * 64-bit as ell.hxx:31 stores them; an id >= cols/block_size marks an empty block), values
 * [rows][ell_cols] row-major.  C (m x n, column-major ldc = m) = alpha*A*B + beta*C with
 * B k x n column-major ldb = k (spmm.hxx:63,67).  fp64 accumulation.
 */
int sm_spmm_bell_f32_ref(const float* values, const uint64_t* column_indices, size_t rows,
                         size_t cols, size_t block_size, size_t ell_cols, const float* B, float* C,
                         size_t n, float alpha, float beta) {
  if (!values || !column_indices || !B || !C || block_size == 0) return SM_ERR_ARG;
  const size_t bcols = ell_cols / block_size;
  const size_t nbc = cols / block_size;
  for (size_t j = 0; j < n; ++j)
    for (size_t i = 0; i < rows; ++i) {
      double acc = 0.0;
      const size_t br = i / block_size;
      for (size_t e = 0; e < bcols; ++e) {
        const uint64_t bc = column_indices[br * bcols + e];
        if (bc >= nbc) continue;
        for (size_t t = 0; t < block_size; ++t)
          acc += (double)values[i * ell_cols + e * block_size + t] *
                 (double)B[j * cols + bc * block_size + t];
      }
      double v = (double)alpha * acc;
      if (beta != 0.0f) v += (double)beta * (double)C[j * rows + i];
      C[j * rows + i] = (float)v;
    }
  return SM_OK;
}

/* COO A (one matrix shared by every batch, spmm.hxx:169 stride 0) times num_batches dense
 * column-major B_b (ldb = B_num_rows, stride B_num_rows*B_num_cols) into C_b (ldc = A_num_rows,
 * stride A_num_rows*B_num_cols).  Duplicate coordinates accumulate.  fp64 accumulation in
 * nnz order per output. */
int sm_spmm_coo_f32_ref(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols,
                        size_t num_batches, const int* rows, const int* colsidx, const float* vals,
                        const float* B, float* C, float alpha, float beta) {
  if (!rows || !colsidx || !vals || !B || !C) return SM_ERR_ARG;
  const size_t bsz = A_num_cols * B_num_cols, csz = A_num_rows * B_num_cols;
  double* acc = (double*)malloc(sizeof(double) * (csz ? csz : 1));
  if (!acc) return SM_ERR_ARG;
  for (size_t b = 0; b < num_batches; ++b) {
    for (size_t i = 0; i < csz; ++i) acc[i] = 0.0;
    for (size_t e = 0; e < A_nnz; ++e) {
      const size_t r = (size_t)rows[e], c = (size_t)colsidx[e];
      if (r >= A_num_rows || c >= A_num_cols) { free(acc); return SM_ERR_ARG; }
      for (size_t j = 0; j < B_num_cols; ++j)
        acc[j * A_num_rows + r] += (double)vals[e] * (double)B[b * bsz + j * A_num_cols + c];
    }
    for (size_t i = 0; i < csz; ++i) {
      double v = (double)alpha * acc[i];
      if (beta != 0.0f) v += (double)beta * (double)C[b * csz + i];
      C[b * csz + i] = (float)v;
    }
  }
  free(acc);
  return SM_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Timed CPU baseline ("port"): the same arithmetic with fp32 accumulation and OpenMP over     */
/* rows, used ONLY by bench.py's cpu_baseline leg.                                            */
/* ------------------------------------------------------------------------------------------ */
/* Dense row-major C = A*B, A given as fp32 (fp16 inputs are widened by the caller's prepass
 * inside the timed region via sm_widen_f16). */
int sm_cpu_gemm_f32(const float* A, const float* B, float* C, size_t m, size_t n, size_t k) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)m; ++i) {
    float* c = C + (size_t)i * n;
    for (size_t j = 0; j < n; ++j) c[j] = 0.0f;
    for (size_t l = 0; l < k; ++l) {
      const float a = A[(size_t)i * k + l];
      const float* brow = B + l * n;
      for (size_t j = 0; j < n; ++j) c[j] += a * brow[j];
    }
  }
  return SM_OK;
}

/* 2:4 path: STRIP-select each strip of fp32 A on the fly (prune + compress fused away) and do
 * only the two kept MACs per strip. */
int sm_cpu_spmma_f32(const float* A, const float* B, float* C, size_t m, size_t n, size_t k) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)m; ++i) {
    float* c = C + (size_t)i * n;
    for (size_t j = 0; j < n; ++j) c[j] = 0.0f;
    for (size_t q = 0; 4 * q < k; ++q) {
      uint32_t key[4];
      float v[4];
      for (unsigned t = 0; t < 4; ++t) {
        v[t] = (4 * q + t < k) ? A[(size_t)i * k + 4 * q + t] : 0.0f;
        uint32_t bits;
        memcpy(&bits, &v[t], 4);
        key[t] = key32(bits);
      }
      const unsigned nib = strip_select(key);
      const unsigned pos[2] = {nib & 3u, nib >> 2};
      for (int t = 0; t < 2; ++t) {
        const size_t kk = 4 * q + pos[t];
        if (kk >= k) continue;
        const float a = v[pos[t]];
        const float* brow = B + kk * n;
        for (size_t j = 0; j < n; ++j) c[j] += a * brow[j];
      }
    }
  }
  return SM_OK;
}

void sm_widen_f16(const uint16_t* in, float* out, size_t count) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)count; ++i) out[i] = h2f(in[i]);
}
/* ------------------------------------------------------------------------------------------ */
/* im2col (extension of the build; follows the unfold + transpose of datasets/get_shapes.py:30-40: */
/* row l = oh * OW + ow of image n, column c * kh * kw + r * kw + u holds                          */
/* X[n][c][oh * stride - pad + r * dil][ow * stride - pad + u * dil], zero outside the image).      */
/* Elements are moved as opaque elt-byte values.                                                    */
/* ------------------------------------------------------------------------------------------ */
int sm_conv_out_size_ref(size_t in, size_t k, size_t stride, size_t pad, size_t dil, size_t* out) {
  if (!out || k == 0 || stride == 0 || dil == 0) return SM_ERR_ARG;
  const size_t span = dil * (k - 1) + 1;
  if (in + 2 * pad < span) return SM_ERR_ARG;
  *out = (in + 2 * pad - span) / stride + 1; /* get_shapes.py:19-20 */
  return SM_OK;
}
int sm_im2col_ref(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                  size_t pad, size_t dil, size_t elt, void* A) {
  size_t OH, OW;
  if (!X || !A || sm_conv_out_size_ref(H, kh, stride, pad, dil, &OH) || sm_conv_out_size_ref(W, kw, stride, pad, dil, &OW))
    return SM_ERR_ARG;
  const size_t K = C * kh * kw, L = OH * OW;
  const unsigned char* x = (const unsigned char*)X;
  unsigned char* a = (unsigned char*)A;
  for (size_t n = 0; n < N; ++n)
    for (size_t oh = 0; oh < OH; ++oh)
      for (size_t ow = 0; ow < OW; ++ow)
        for (size_t c = 0; c < C; ++c)
          for (size_t r = 0; r < kh; ++r)
            for (size_t u = 0; u < kw; ++u) {
              const long ih = (long)(oh * stride + r * dil) - (long)pad, iw = (long)(ow * stride + u * dil) - (long)pad;
              unsigned char* dst = a + ((n * L + oh * OW + ow) * K + (c * kh + r) * kw + u) * elt;
              if (ih >= 0 && ih < (long)H && iw >= 0 && iw < (long)W)
                memcpy(dst, x + (((n * C + c) * H + (size_t)ih) * W + (size_t)iw) * elt, elt);
              else
                memset(dst, 0, elt);
            }
  return SM_OK;
}

void sm_widen_bf16(const uint16_t* in, float* out, size_t count) {
  for (size_t i = 0; i < count; ++i) out[i] = bf2f(in[i]);
}
void sm_narrow_bf16(const float* in, uint16_t* out, size_t count) {
  for (size_t i = 0; i < count; ++i) out[i] = f2bf(in[i]);
}
void sm_narrow_f16(const float* in, uint16_t* out, size_t count) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)count; ++i) out[i] = f2h(in[i]);
}

int sm_oracle_num_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
