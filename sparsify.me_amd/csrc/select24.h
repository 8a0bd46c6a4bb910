// select24.h -- the frozen 2:4 STRIP selection rule as device functions (shared by prune.hip and the
// fused prune+matmul kernel).  Mirrors oracle/sm_oracle.c: strip_select.
#pragma once
#include "sm_common.h"

namespace sm {

// ---------------------------------------------------------------------------------------------
// selection rules (mirror oracle/sm_oracle.c: strip_select, tile_select)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t key_of(uint16_t v) { return v & 0x7fffu; }
__device__ __forceinline__ uint32_t key_of(uint32_t v) { return v & 0x7fffffffu; }

// 4-bit keep mask (bit t = position t kept) of the STRIP rule: top-2 keys, ties -> lower index.
__device__ __forceinline__ unsigned strip_keepmask(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
  const unsigned r0 = (k1 > k0) + (k2 > k0) + (k3 > k0);
  const unsigned r1 = (k0 >= k1) + (k2 > k1) + (k3 > k1);
  const unsigned r2 = (k0 >= k2) + (k1 >= k2) + (k3 > k2);
  const unsigned r3 = (k0 >= k3) + (k1 >= k3) + (k2 >= k3);
  return (r0 < 2 ? 1u : 0u) | (r1 < 2 ? 2u : 0u) | (r2 < 2 ? 4u : 0u) | (r3 < 2 ? 8u : 0u);
}
// keep mask (exactly two bits set) -> metadata nibble p0 | p1 << 2.
__device__ __forceinline__ unsigned nibble_of(unsigned keep) {
  const unsigned p0 = __builtin_ctz(keep);
  const unsigned p1 = 31u - __builtin_clz(keep);
  return p0 | (p1 << 2);
}

// fp16 strip in one pass (23 VALU ops against ~60 for the mask form above): element i of the strip becomes the
// 32-bit composite key (|x_i| << 16) | (3 - i) -- all four distinct, larger = kept earlier, equal magnitudes
// ordered by the lower index -- and the two largest fall out of five instructions:
//   m = max3(K0,K1,K2)   med = med3(K0,K1,K2)   first = max(m,K3)   second = max(min(m,K3), med).
// Their low two bits name the kept positions; one v_perm_b32 with a computed selector then pulls the two kept
// halves (sign and all) out of the strip's two dwords in position order.
//   d0 = {x1:x0}, d1 = {x3:x2}  ->  kept = {x[p1]:x[p0]},  nib = p0 | p1 << 2   (p0 < p1)
// Same result as strip_keepmask/nibble_of for every input (tests/test_gpu_parity.py: compress vs oracle,
// fused vs staged, incl. ties, +-0, inf, NaN patterns).
__device__ __forceinline__ void strip_select_f16(uint32_t d0, uint32_t d1, uint32_t& kept, uint32_t& nib) {
  const uint32_t a0 = d0 & 0x7fff7fffu, a1 = d1 & 0x7fff7fffu;
  const uint32_t K0 = (a0 << 16) | 3u, K1 = (a0 & 0xffff0000u) | 2u;
  const uint32_t K2 = (a1 << 16) | 1u, K3 = a1 & 0xffff0000u;
  // first = max of the four; second = max(min(m, K3), med3(K0, K1, K2)) with m = max3(K0, K1, K2): if m >= K3 the
  // runner-up is K3 or the median of the three, otherwise it is m itself (>= that median).  Five instructions.
  const uint32_t m01 = K0 > K1 ? K0 : K1, n01 = K0 > K1 ? K1 : K0;
  const uint32_t m = m01 > K2 ? m01 : K2;          // max(max(K0,K1),K2): v_max3_u32
  const uint32_t c01 = m01 < K2 ? m01 : K2;        // min(max(K0,K1),K2)
  const uint32_t med = n01 > c01 ? n01 : c01;      // max(min(K0,K1), min(max(K0,K1),K2)): v_med3_u32
  const uint32_t first = m > K3 ? m : K3, lo = m > K3 ? K3 : m;
  const uint32_t second = lo > med ? lo : med;
  const uint32_t a = first & 3u, b = second & 3u;       // 3 - position
  const uint32_t A = a > b ? a : b, B = a > b ? b : a;  // p0 = 3 - A < p1 = 3 - B
  const uint32_t sel = 0x07060706u - (A | (B << 16)) * 0x0202u;  // < 2^24: v_mul_u32_u24
  kept = __builtin_amdgcn_perm(d1, d0, sel);
  nib = 15u - (A | (B << 2));
}

// TILE rule: 16-bit keep mask (bit 4*r + c) of the best of the 90 candidates.
__device__ __forceinline__ unsigned tile_keepmask(const float (&mag)[4][4]) {
  float s0[6], s1[6], s2[6], s3[6];
#define SM_PAIRS(S, R)            \
  S[0] = mag[R][0] + mag[R][1];   \
  S[1] = mag[R][0] + mag[R][2];   \
  S[2] = mag[R][0] + mag[R][3];   \
  S[3] = mag[R][1] + mag[R][2];   \
  S[4] = mag[R][1] + mag[R][3];   \
  S[5] = mag[R][2] + mag[R][3];
  SM_PAIRS(s0, 0) SM_PAIRS(s1, 1) SM_PAIRS(s2, 2) SM_PAIRS(s3, 3)
#undef SM_PAIRS
  float best = -1.0f;
  unsigned bm = 0;
#define TILE_CAND(I, P0, P1, P2, P3, MK)                        \
  {                                                             \
    const float sc = (s0[P0] + s1[P1]) + (s2[P2] + s3[P3]);     \
    if (sc > best) {                                            \
      best = sc;                                                \
      bm = MK;                                                  \
    }                                                           \
  }
#include "tile_patterns.inc"
#undef TILE_CAND
  return bm;
}


}  // namespace sm
