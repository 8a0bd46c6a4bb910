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

}  // namespace sm
