// mma_tile.h -- shared device pieces of the fp16 matrix kernels (dense MFMA GEMM and 2:4 SMFMAC):
// LDS images, swizzles and fragment reads.  gfx950 only.
//
// Operand maps (verified on hardware with exact integer data, profiles/probe_gfx950_r01.txt):
//   v_mfma_f32_16x16x32_f16   A lane l: row l&15, k = 8*(l>>4)+j (j<8);  B lane l: col l&15, same k;
//                             D lane l: col l&15, rows 4*(l>>4)+r (r<4).
//   v_smfmac_f32_16x16x64_f16 A lane l: row l&15, 8 compressed values = the kept pairs of the four
//                             strips covering dense k = 16*(l>>4) .. +15; index bits [2e+1:2e] of the
//                             low (abid 0) / high (abid 1) 16 bits give slot e's position in its strip;
//                             B lane l: col l&15, element j<8 is k = 8*(l>>4)+j, element j>=8 is
//                             k = 32 + 8*(l>>4) + (j-8);  D as above.
//   ds_read_b64_tr_b16        per 16-lane group: lane 4q+p supplies &blk[q][4p] (4 rows x 16 cols of
//                             16-bit); lane i receives {blk[0][i], blk[1][i], blk[2][i], blk[3][i]}.
#pragma once
#include "sm_common.h"

namespace sm {

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) s4 lds_s4;

// ---- A-side image: [rows][128 B] (64 halves of one row per LDS row), 16-byte chunk c of row r
//      stored at chunk c ^ (r & 7): conflict-free for the ds_read_b128 fragment reads of both
//      instruction families (16 rows x one chunk column per 16-lane access group).
__device__ __forceinline__ unsigned a_off(unsigned row, unsigned chunk) {
  return row * 128u + 16u * (chunk ^ (row & 7u));
}

// ---- B-side image: panels of 64 columns; panel p at p * (KROWS * 128); row = k index, 128 B per
//      row; chunk c of k-row kr stored at chunk c ^ fB(kr), fB from bits 1 and 3 of kr: the two
//      4-row blocks a 32-lane half fetches with one ds_read_b64_tr_b16 (rows o..o+3 and o+8..o+11)
//      then cover all 64 banks exactly once.
__device__ __forceinline__ unsigned b_swz(unsigned kr) { return (((kr >> 1) & 1u) << 1) | (((kr >> 3) & 1u) << 2); }
template <int KROWS>
__device__ __forceinline__ unsigned b_off(unsigned kr, unsigned col /*0..BN-1, multiple of 4*/) {
  const unsigned panel = col >> 6, cl = col & 63u;
  return panel * (KROWS * 128u) + kr * 128u + 16u * ((cl >> 3) ^ b_swz(kr)) + 2u * (cl & 7u);
}

// Transposed fragment read: this lane's 4 k-values (rows kr0 + q .. as the group map says) of
// column col0 + (lane & 15).  kr0 = first row of the 16-lane group's 4x16 block.
template <int KROWS>
__device__ __forceinline__ s4 b_read_tr(const char* Bs, unsigned kr0, unsigned col0, unsigned lane) {
  const unsigned i = lane & 15u, q = i >> 2, p = i & 3u;
  const char* addr = Bs + b_off<KROWS>(kr0 + q, col0 + 4u * p);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds_char*)addr);
}

// XCD-aware bijective remap of a linear workgroup id: blocks that share an XCD (ids equal mod 8)
// receive a contiguous range of logical ids, so tiles that re-read the same operand panel sit on
// one L2 (speed only; any placement is correct).
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, x = id & 7u;
  const unsigned base = x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q;
  return base + (id >> 3);
}

// ---- LDS-DMA (global_load_lds) plumbing shared by the pipelined kernels
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// s_waitcnt vmcnt(N) + s_barrier as ONE statement: the DMA of the stage about to be read has landed
// for every wave, and every wave has finished reading the buffer about to be refilled.  (A plain
// __syncthreads() would drain vmcnt to 0 and serialise the ring.)
template <int N>
__device__ __forceinline__ void wait_dma_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

}  // namespace sm
