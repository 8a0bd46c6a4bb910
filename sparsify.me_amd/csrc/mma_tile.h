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
#include "select24.h"

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

// 64-byte-row A image: 16-byte chunk c of row r lives at chunk c ^ ((-(r >> 2)) & 3).
__device__ __forceinline__ unsigned a64_swz(unsigned row) { return (0u - (row >> 2)) & 3u; }

// ---- the two 16-bit element types.  Kernels keep `half_t` as the 16-bit storage type and take `bool BF`: false =
//      IEEE fp16, true = bfloat16 (same blob, same selection -- it only looks at the magnitude bits -- other matrix
//      instruction and other rounding at the end).  Conversions round to nearest even, once.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
template <bool BF>
__device__ __forceinline__ half_t to_elt(float x) {
  if constexpr (BF) return __builtin_bit_cast(half_t, (__bf16)x);  // v_cvt_pk_bf16_f32
  else return (half_t)x;
}
template <bool BF>
__device__ __forceinline__ float to_f32(half_t x) {
  if constexpr (BF) return __builtin_bit_cast(float, (uint32_t)__builtin_bit_cast(unsigned short, x) << 16);
  else return (float)x;
}
template <bool BF>
__device__ __forceinline__ f4 smfmac16(h8 a, h16 b, f4 c, int idx) {
  if constexpr (BF) return __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf16x16, b), c, idx, 0, 0);
  else return __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, idx, 0, 0);
}
template <bool BF>
__device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// One 64-deep stage of the 2:4 matmul for one consumer wave (wave tile 16*FM x 16*FN at rows row0.., columns col0..
// of the workgroup tile): A fragments + index halfwords from the 64-byte-row image `As` / its metadata `Ms`, B
// fragments from the row-major [64][BN] image `Bs` through ds_read_b64_tr_b16, FM x FN v_smfmac_f32_16x16x64_f16.
// The transposing reads are issued by hand (inline asm, counted lgkmcnt): the compiler would drain the in-flight DMA
// (vmcnt(0)) in front of the intrinsic form; fragment j+1's four reads are in flight while fragment j's SMFMACs run.
template <int FM, int FN, bool BF = false>
__device__ __forceinline__ void smfmac_stage(const char* As, const char* Ms, const char* Bs, unsigned row0, unsigned col0,
                                             unsigned lane, f4 (&acc)[FM][FN]) {
  const unsigned g = lane >> 4, r = lane & 15u;
  h8 af[FM];
  int idx[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const unsigned row = row0 + i * 16 + r;
    af[i] = *reinterpret_cast<const h8*>(As + row * 64u + 16u * (g ^ a64_swz(row)));
    idx[i] = (int)*reinterpret_cast<const unsigned short*>(Ms + row * 8u + 2u * g);
  }
  const unsigned bs_addr = (unsigned)(uintptr_t)(lds_char*)Bs;
  s4 t0[2], t1[2], t2[2], t3[2];
  auto issue = [&](int j, s4& v0, s4& v1, s4& v2, s4& v3) {
    const unsigned c0 = col0 + j * 16, q = r >> 2, pp = r & 3u;
    const unsigned a = bs_addr + b_off<64>(8u * g + q, c0 + 4u * pp);
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:4096\n\tds_read_b64_tr_b16 %3, %4 offset:4608"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a) : "memory");
  };
  issue(0, t0[0], t1[0], t2[0], t3[0]);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c = j & 1, n = c ^ 1;
    if (j + 1 < FN) {
      issue(j + 1, t0[n], t1[n], t2[n], t3[n]);
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(t0[c]), "+v"(t1[c]), "+v"(t2[c]), "+v"(t3[c]) :: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0[c]), "+v"(t1[c]), "+v"(t2[c]), "+v"(t3[c]) :: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    typedef short s16 __attribute__((ext_vector_type(16)));
    const s16 all = {t0[c][0], t0[c][1], t0[c][2], t0[c][3], t1[c][0], t1[c][1], t1[c][2], t1[c][3],
                     t2[c][0], t2[c][1], t2[c][2], t2[c][3], t3[c][0], t3[c][1], t3[c][2], t3[c][3]};
    const h16 bf = __builtin_bit_cast(h16, all);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][j] = smfmac16<BF>(af[i], bf, acc[i][j], idx[i]);
  }
}

// B sweep of one 64-deep stage for one wave: B fragments from the row-major [64][BN] image `Bs` through
// ds_read_b64_tr_b16 (issued by hand, counted lgkmcnt: fragment j+1's four reads in flight while fragment j's SMFMACs run),
// FM x FN v_smfmac_f32_16x16x64 with the A operands / index halfwords the caller built.
template <int FM, int FN, bool BF = false, int D = 1>
__device__ __forceinline__ void smfmac_b_sweep(const h8 (&af)[FM], const int (&idx)[FM], const char* Bs, unsigned col0, unsigned lane,
                                               f4 (&acc)[FM][FN]) {
  // D = B fragments requested ahead of the one being multiplied (1: the default; 2: two fragments = eight reads in flight,
  // for wave tiles with few SMFMACs per fragment, where one fragment's two SMFMACs do not cover the next one's LDS latency)
  static_assert(D == 1 || D == 2, "fragments in flight");
  const unsigned g = lane >> 4, r = lane & 15u;
  const unsigned bs_addr = (unsigned)(uintptr_t)(lds_char*)Bs;
  s4 t0[D + 1], t1[D + 1], t2[D + 1], t3[D + 1];
  auto issue = [&](int j, s4& v0, s4& v1, s4& v2, s4& v3) {
    const unsigned c0 = col0 + j * 16, q = r >> 2, pp = r & 3u;
    const unsigned a = bs_addr + b_off<64>(8u * g + q, c0 + 4u * pp);
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:4096\n\tds_read_b64_tr_b16 %3, %4 offset:4608"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a) : "memory");
  };
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (j < FN) issue(j, t0[j], t1[j], t2[j], t3[j]);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c = j % (D + 1), n = (j + D) % (D + 1);
    if (j + D < FN) {
      issue(j + D, t0[n], t1[n], t2[n], t3[n]);
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(t0[c]), "+v"(t1[c]), "+v"(t2[c]), "+v"(t3[c]) : "n"(4 * D) : "memory");
    } else if (D == 2 && j + 1 < FN) {
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(t0[c]), "+v"(t1[c]), "+v"(t2[c]), "+v"(t3[c]) :: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0[c]), "+v"(t1[c]), "+v"(t2[c]), "+v"(t3[c]) :: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    typedef short s16 __attribute__((ext_vector_type(16)));
    const s16 all = {t0[c][0], t0[c][1], t0[c][2], t0[c][3], t1[c][0], t1[c][1], t1[c][2], t1[c][3],
                     t2[c][0], t2[c][1], t2[c][2], t2[c][3], t3[c][0], t3[c][1], t3[c][2], t3[c][3]};
    const h16 bf = __builtin_bit_cast(h16, all);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][j] = smfmac16<BF>(af[i], bf, acc[i][j], idx[i]);
  }
}

// 16 dense halves of one row (k = 16 g .. 16 g + 15 of the stage, as two 16-byte vectors) -> the lane's SMFMAC A operand
// (four kept pairs) and index halfword: four strip selections (select24.h).
__device__ __forceinline__ void dense16_to_operand(const u4 lo, const u4 hi, h8& af, int& idx) {
  uint32_t k0, k1, k2, k3, n0, n1, n2, n3;
  strip_select_f16(lo[0], lo[1], k0, n0);
  strip_select_f16(lo[2], lo[3], k1, n1);
  strip_select_f16(hi[0], hi[1], k2, n2);
  strip_select_f16(hi[2], hi[3], k3, n3);
  af = __builtin_bit_cast(h8, u4{k0, k1, k2, k3});
  idx = (int)(n0 | (n1 << 4) | (n2 << 8) | (n3 << 12));
}

// The stage straight from the DENSE A: `Araw` is the [rows][128 B] image of 64 dense k per row (chunk c of row r
// at chunk c ^ (r & 7), a_off), and the lane that would read 8 compressed halves + 4 nibbles reads its 16 dense halves
// (two ds_read_b128) and selects in registers: four strips -> the A operand and the index halfword of
// v_smfmac_f32_16x16x64_f16.  No compressed image, no metadata, no selecting loader waves.
template <int FM, int FN, bool BF = false, int D = 1>
__device__ __forceinline__ void smfmac_stage_dense_a(const char* Araw, const char* Bs, unsigned row0, unsigned col0,
                                                     unsigned lane, f4 (&acc)[FM][FN]) {
  const unsigned g = lane >> 4, r = lane & 15u;
  h8 af[FM];
  int idx[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const unsigned row = row0 + i * 16 + r;
    const u4 lo = *reinterpret_cast<const u4*>(Araw + a_off(row, 2u * g));
    const u4 hi = *reinterpret_cast<const u4*>(Araw + a_off(row, 2u * g + 1u));
    dense16_to_operand(lo, hi, af[i], idx[i]);
  }
  smfmac_b_sweep<FM, FN, BF, D>(af, idx, Bs, col0, lane, acc);
}

// ---- the DENSE twin of the two functions above (round 4): the same stage images, the same B reads, dense v_mfma_f32_16x16x32 on
// ALL of A's elements instead of the 2:4 selection + v_smfmac -- so that the dense GEMM the 2:4 path is measured against runs
// through the very pipelines (direct / big / span) the fused kernels use.  A B fragment's 16 halves are already two dense
// operands (elements 0-7: k = 8 g + j of the stage's first 32-k block, 8-15: the second block's); the A lane takes the matching
// 8 + 8 halves of its row: chunks g and 4 + g of the 128-byte row image.
template <int FM, int FN, bool BF = false>
__device__ __forceinline__ void mfma_b_sweep(const h8 (&a0)[FM], const h8 (&a1)[FM], const char* Bs, unsigned col0, unsigned lane, f4 (&acc)[FM][FN]) {
  const unsigned g = lane >> 4, r = lane & 15u;
  const unsigned bs_addr = (unsigned)(uintptr_t)(lds_char*)Bs;
  s4 t0[2], t1[2], t2[2], t3[2];
  auto issue = [&](int j, s4& v0, s4& v1, s4& v2, s4& v3) {
    const unsigned c0 = col0 + j * 16, q = r >> 2, pp = r & 3u;
    const unsigned a = bs_addr + b_off<64>(8u * g + q, c0 + 4u * pp);
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:4096\n\tds_read_b64_tr_b16 %3, %4 offset:4608"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a) : "memory");
  };
  issue(0, t0[0], t1[0], t2[0], t3[0]);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c = j & 1, n = c ^ 1;
    if (j + 1 < FN) {
      issue(j + 1, t0[n], t1[n], t2[n], t3[n]);
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(t0[c]), "+v"(t1[c]), "+v"(t2[c]), "+v"(t3[c]) :: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0[c]), "+v"(t1[c]), "+v"(t2[c]), "+v"(t3[c]) :: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    typedef short s8v __attribute__((ext_vector_type(8)));
    const s8v lo = {t0[c][0], t0[c][1], t0[c][2], t0[c][3], t1[c][0], t1[c][1], t1[c][2], t1[c][3]};
    const s8v hi = {t2[c][0], t2[c][1], t2[c][2], t2[c][3], t3[c][0], t3[c][1], t3[c][2], t3[c][3]};
    const h8 b0 = __builtin_bit_cast(h8, lo), b1 = __builtin_bit_cast(h8, hi);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<BF>(a0[i], b0, acc[i][j]);
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<BF>(a1[i], b1, acc[i][j]);
  }
}
template <int FM, int FN, bool BF = false>
__device__ __forceinline__ void mfma_stage_dense_a(const char* Araw, const char* Bs, unsigned row0, unsigned col0, unsigned lane, f4 (&acc)[FM][FN]) {
  const unsigned g = lane >> 4, r = lane & 15u;
  h8 a0[FM], a1[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const unsigned row = row0 + i * 16 + r;
    a0[i] = *reinterpret_cast<const h8*>(Araw + a_off(row, g));
    a1[i] = *reinterpret_cast<const h8*>(Araw + a_off(row, 4u + g));
  }
  mfma_b_sweep<FM, FN, BF>(a0, a1, Bs, col0, lane, acc);
}

// Epilogue of the 2:4 matmul kernels, called by EVERY thread of the workgroup after its last barrier: the SMFMAC
// result map leaves 4 consecutive ROWS of one column per lane, so with beta == 0 and a 16-byte aligned C the tile is
// transposed through LDS (`smem`, BM x (2 BN + 16) bytes, aliasing the stage buffers) and written as 16-byte row
// pieces by all NT threads; otherwise (beta != 0, misaligned C) each accumulator wave stores its elements itself,
// reading C once: one rounding of alpha * acc + beta * C to fp16 either way.  `has_acc`: this wave holds a tile
// (loader waves of the producer/consumer kernels do not); row0/col0: its tile inside the workgroup tile.
template <int BM, int BN, int FM, int FN, int NT, bool BF = false>
__device__ __forceinline__ void store_c_tile(char* smem, half_t* C, const f4 (&acc)[FM][FN], bool has_acc, unsigned row0,
                                             unsigned col0, int m0, int n0, int Mrows, int N, float alpha, float beta,
                                             unsigned tid) {
  constexpr int CPITCH = BN * 2 + 16;
  const unsigned lane = tid & 63u, g = lane >> 4, r = lane & 15u;
  const bool c_vec = (reinterpret_cast<uintptr_t>(C) & 15u) == 0 && (N % 8 == 0);
  if (beta == 0.0f && c_vec) {
    if (has_acc) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const unsigned row = row0 + i * 16 + 4u * g, col = col0 + j * 16 + r;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<half_t*>(smem + (row + q) * CPITCH + col * 2) = to_elt<BF>(alpha * acc[i][j][q]);
        }
    }
    __syncthreads();
    constexpr int NCH = BM * (BN / 8);
    for (unsigned q = tid; q < (unsigned)NCH; q += (unsigned)NT) {
      const unsigned row = q / (BN / 8), cn = q % (BN / 8);
      const int gr = m0 + (int)row, gc = n0 + 8 * (int)cn;
      if (gr >= Mrows || gc >= N) continue;  // N % 8 == 0: a chunk is all in or all out
      // non-temporal: C is written once and not read again by this kernel; keeping it out of the way of the operands in
      // L2 / the Infinity Cache is worth 11 % on the 2:4 matmul of the ResNet-50 table (1.67 -> 1.48 ms, cold buffers)
      __builtin_nontemporal_store(*reinterpret_cast<const u4*>(smem + row * CPITCH + cn * 16), reinterpret_cast<u4*>(C + (size_t)gr * N + gc));
    }
  } else if (has_acc) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int gc = n0 + (int)(col0 + j * 16 + r);
        if (gc >= N) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gr = m0 + (int)(row0 + i * 16 + 4u * g) + q;
          if (gr >= Mrows) continue;
          half_t* dst = C + (size_t)gr * N + gc;
          float v = alpha * acc[i][j][q];
          if (beta != 0.0f) v += beta * to_f32<BF>(*dst);
          *dst = to_elt<BF>(v);
        }
      }
  }
}

// XCD-aware bijective remap of a linear workgroup id: blocks that share an XCD (ids equal mod 8)
// receive a contiguous range of logical ids, so tiles that re-read the same operand panel sit on
// one L2 (speed only; any placement is correct).
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, x = id & 7u;
  const unsigned base = x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q;
  return base + (id >> 3);
}

// One column tile per row tile: no operand panel is shared between workgroups (B is small and L2-resident everywhere), and the workgroups keep the
// DISPATCH order -- the chip then reads one moving window of A instead of eight (one per XCD), which is what the fastest device copy does
// (tools/probes/copy_probe.hip).  Measured on the direct fused kernel (profiles/ab_remap_r05a{t,u}.txt, same C bit for bit): 12544 x 64 x 576 x 3
// 270.7 -> 252.4 us, 12544 x 64 x 256 x 2 93.5 -> 88.6, the others 0-1.5 % faster; on the kernels whose column tiles share A rows or whose B tiles
// are large (big / wide / A-stationary) the XCD ranges stay: 0-5 % slower without them.
__device__ __forceinline__ unsigned tile_order(unsigned id, unsigned nwg, bool dispatch_order) { return dispatch_order ? id : xcd_remap(id, nwg); }

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
