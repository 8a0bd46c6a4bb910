// spmma_f32_split.hip -- the fp32 2:4 product on the SPARSE matrix instruction (round 4): C = prune24_strip(A) * B for fp32
// operands, computed by v_smfmac_f32_16x16x64_bf16 on exact bfloat16 splits of both operands.
//
// gfx950 has no fp32 sparse matrix instruction: sm_spmma_fused_f32 (gemm_f32.hip) expands the 2:4 operand back to dense
// v_mfma_f32_16x16x4_f32 work -- 157 TF/s peak, the dense fp32 GEMM's rate at best, never the 2 x a 2:4 operand promises.
// Here every fp32 value x is split EXACTLY into three bfloat16 pieces x = x1 + x2 + x3 (the top 8, middle 8 and low 8 bits
// of its 24-bit significand: three truncations, each residual computed exactly in fp32), and
//   a * b  ~  a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1        (planes = 3: dropped terms a2 b3 + a3 b2 + a3 b3 < 2^-21 |a b|)
//   a * b  ~  a1 b1 + a1 b2 + a2 b1                                  (planes = 2: |error| < 2^-13 |a b|)
// each product exact in the instruction's fp32 accumulator arithmetic.  Six (three) sparse instructions per 16 x 16 x 64
// block at ~3.5 PF/s dense-equivalent are 0.6 (1.2) PF/s of fp32-equivalent work against 0.157: the path becomes what the
// fp16 path is, bound by the HBM stream of A.  The 2:4 selection itself is made on the fp32 values (frozen STRIP rule,
// select24.h / oracle strip_select), so the mask is the exact path's bit for bit; what differs from sm_spmma_fused_f32 is
// the rounding of the products (stated bound, asserted in tests/test_gpu_parity.py::test_spmma_f32_split_*), inside
// north_star's 1e-3 for fp32 by three (planes = 2) to six (planes = 3) orders of magnitude.  Reference: the matmul step of
// sparsifyme::spmma<float>, include/sparsify.me/spmma.hxx:106-114 (cuSPARSELt computes fp32 operands in TF32: 10 bits).
//
// Structure = the fp16 direct fused kernel in fp32 bytes: 128-row tiles, eight waves (wave w owns rows 16 w .. 16 w + 15
// and all BN columns: every strip is selected once, by the lane that feeds it), the DENSE fp32 A stage (128 rows x 256 B)
// and the bfloat16 planes of the B stage (planes x 64 x BN) by LDS-DMA into a ring of two, one counted wait + barrier per
// stage.  B is split once per call by a streaming pre-pass into the caller's workspace (planes x k x n bfloat16).
// Non-finite operand values: the residual of an inf / NaN is NaN, so every output such a value reaches is NaN (the exact
// path may say +-inf there); all other outputs are untouched (tests: test_spmma_f32_split_edges).
#include <algorithm>

#include "select24.h"
#include "spmma_args.h"

namespace sm {

struct SplitArgs {
  const float* A;
  const unsigned short* Bp;  // [plane][batchB][k][n] bfloat16
  float* C;
  size_t sA, sBp, plane, sC;  // elements
  int Mrows, N, K, lda;
  int batch, tiles_m, tiles_n;
  float alpha, beta;
  int xcd_ranges;  // tuning builds (SM_SPLIT_XCD=1): the XCD ranges of round 4 instead of the dispatch order for single-column-tile launches
};

__device__ __forceinline__ float as_f32(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ uint32_t as_u32(float x) { return __builtin_bit_cast(uint32_t, x); }

// residual of a truncation to bfloat16: exact (a non-finite x gives NaN: see the header)
__device__ __forceinline__ float trunc_residual(uint32_t x) { return as_f32(x) - as_f32(x & 0xffff0000u); }

// {hi16(b) : hi16(a)}: two bfloat16 (truncated) in SMFMAC operand order
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// One lane's 16 dense fp32 of a row (k = 16 g .. 16 g + 15 of the stage) -> NP bfloat16 SMFMAC A operands + the index halfword.
// The frozen STRIP rule (top-2 magnitudes, ties to the lower index; oracle strip_select) as a 5-compare selection network on
// the 31-bit magnitude keys: pair winners (c01, c23), winner of winners (cw), then the runner-up among the three left (ca:
// loser of {0, 1} against the winner of {2, 3}; cb: winner of {0, 1} against the loser of {2, 3}).  Which positions are kept
// is then pure boolean algebra on the five compare results -- lane masks in scalar registers, combined on the scalar unit --
// and the kept values / their positions fall out of two conditional moves each: ~23 vector instructions per strip against
// ~45 for the compare-and-count form with integer masks (strip_keepmask), same mask for every input
// (tests: test_spmma_f32_split on tie-heavy integer data must equal the exact kernel bit for bit).
template <int NP>
__device__ __forceinline__ void dense16_f32_to_operands(const u4 (&v)[4], h8 (&af)[NP], int& idx) {
  uint32_t pk[NP][4];
  unsigned meta = 0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t x0 = v[s][0], x1 = v[s][1], x2 = v[s][2], x3 = v[s][3];
    const uint32_t k0 = x0 & 0x7fffffffu, k1 = x1 & 0x7fffffffu, k2 = x2 & 0x7fffffffu, k3 = x3 & 0x7fffffffu;
    // five compares -> lane masks in scalar registers (ballot); the keep logic is scalar boolean algebra on them, and the
    // masks return as conditions of the conditional moves (inverse ballot).  Written with the builtins on purpose: as plain
    // bool expressions the compiler either branched around every strip (&&, ||, ?:) or materialised the bools in vector registers.
    typedef unsigned long long lm_t;
    const lm_t c01 = __builtin_amdgcn_ballot_w64(k0 >= k1), c23 = __builtin_amdgcn_ballot_w64(k2 >= k3);
    const uint32_t w01 = k0 > k1 ? k0 : k1, l01 = k0 > k1 ? k1 : k0, w23 = k2 > k3 ? k2 : k3, l23 = k2 > k3 ? k3 : k2;  // v_max_u32 / v_min_u32
    const lm_t cw = __builtin_amdgcn_ballot_w64(w01 >= w23), ca = __builtin_amdgcn_ballot_w64(l01 >= w23), cb = __builtin_amdgcn_ballot_w64(w01 >= l23);
    const bool keep0 = __builtin_amdgcn_inverse_ballot_w64((cw & (c01 | ca)) | (~cw & cb & c01));
    const bool keep1 = __builtin_amdgcn_inverse_ballot_w64((cw & (~c01 | ca)) | (~cw & cb & ~c01));
    const bool keep2 = __builtin_amdgcn_inverse_ballot_w64((cw & ~ca & c23) | (~cw & (c23 | ~cb)));
    const bool keep3 = __builtin_amdgcn_inverse_ballot_w64((cw & ~ca & ~c23) | (~cw & (~c23 | ~cb)));
    const uint32_t lo = keep0 ? x0 : (keep1 ? x1 : x2);  // the kept value at the lower position p0 ...
    const uint32_t hi = keep3 ? x3 : (keep2 ? x2 : x1);  // ... and at the higher position p1
    const unsigned q0 = keep0 ? 0u : (keep1 ? (1u << (4 * s)) : (2u << (4 * s)));             // p0 << 4 s
    const unsigned q1 = keep3 ? (12u << (4 * s)) : (keep2 ? (8u << (4 * s)) : (4u << (4 * s)));  // p1 << (4 s + 2)
    meta |= q0 | q1;
    pk[0][s] = pack_hi16(lo, hi);
    if constexpr (NP >= 2) {
      const float rl = trunc_residual(lo), rh = trunc_residual(hi);
      pk[1][s] = pack_hi16(as_u32(rl), as_u32(rh));
      if constexpr (NP >= 3) pk[2][s] = pack_hi16(as_u32(trunc_residual(as_u32(rl))), as_u32(trunc_residual(as_u32(rh))));  // <= 8 significant bits left: exact
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) af[p] = __builtin_bit_cast(h8, u4{pk[p][0], pk[p][1], pk[p][2], pk[p][3]});
  idx = (int)meta;
}

// DENSE (sm_gemm_rowmajor_f32_split): no selection -- every element of A is split and multiplied, by v_mfma_f32_16x16x32_bf16 (a B
// fragment's 16 values are already two dense operands: the stage's k = 8 g + j and 32 + 8 g + j): the dense product by the same
// pieces, the comparator the 2:4 form should be held against.
template <int BN, int NP, bool ANT, int NW, bool DENSE = false>
__global__ __launch_bounds__(64 * NW) void spmma_f32_split_kernel(const SplitArgs p) {
  constexpr int BM = 128, TM = BM / NW, FM = TM / 16, FN = BN / 16;
  constexpr int SA = BM * 256, SBP = 64 * BN * 2, STAGE = SA + NP * SBP;
  constexpr int A_N = BM / 4, B_N = NP * (BN / 8);  // 1 KiB DMA wave-instructions per stage
  static_assert(A_N % NW == 0 && B_N % NW == 0, "equal DMA share per wave");
  constexpr int SLA = A_N / NW, SLB = B_N / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, p.tiles_n == 1 && !p.xcd_ranges);  // (mma_tile.h)
  const unsigned b = lid / tiles, trem = lid - b * tiles;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = p.K / 64;
  const float* A = p.A + (size_t)b * p.sA;
  const unsigned short* Bp = p.Bp + (size_t)b * p.sBp;
  float* C = p.C + (size_t)b * p.sC;
  const int mlast = p.Mrows - 1;

  const char* asrc[SLA];
  unsigned aoff[SLA];
  const char* bsrc[SLB];
  unsigned boff[SLB];
#pragma unroll
  for (int i = 0; i < SLA; ++i) {  // 4 rows x 256 B: lane -> row 4 t + lane / 16, LDS chunk lane % 16 holds source chunk (lane % 16) ^ (row & 15)
    const unsigned t = wave + (unsigned)NW * i;
    const unsigned row = 4u * t + (lane >> 4), cs = (lane & 15u) ^ (row & 15u);
    int gr = m0 + (int)row;
    gr = gr < mlast ? gr : mlast;
    asrc[i] = reinterpret_cast<const char*>(A + (size_t)gr * p.lda) + 16u * cs;
    aoff[i] = t * 1024u;
  }
#pragma unroll
  for (int i = 0; i < SLB; ++i) {  // plane pl, 8 k-rows x 128 B of one 64-column panel (the fp16 kernels' B image, per plane)
    const unsigned t = wave + (unsigned)NW * i;
    const unsigned pl = t / (unsigned)(BN / 8), j = t - pl * (unsigned)(BN / 8);
    const unsigned panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
    const unsigned cs = (lane & 7u) ^ b_swz(kr);
    int gc = n0 + (int)(64u * panel + 8u * cs);
    gc = gc <= p.N - 8 ? gc : p.N - 8;
    bsrc[i] = reinterpret_cast<const char*>(Bp + (size_t)pl * p.plane + (size_t)kr * p.N + gc);
    boff[i] = SA + pl * SBP + panel * 8192u + (j & 7u) * 1024u;
  }
  const size_t bstep = (size_t)64 * p.N * 2;
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < SLB; ++i) __builtin_amdgcn_global_load_lds((gptr_t*)(bsrc[i] + (size_t)kt * bstep), (lptr_t*)(base + boff[i]), 16, 0, 0);
    // A is read once by the whole grid when there is one column tile: non-temporal (ANT); otherwise its re-reads come from L2
#pragma unroll
    for (int i = 0; i < SLA; ++i) {
      if (ANT) __builtin_amdgcn_global_load_lds((gptr_t*)(asrc[i] + (size_t)kt * 256), (lptr_t*)(base + aoff[i]), 16, 0, 2);
      else __builtin_amdgcn_global_load_lds((gptr_t*)(asrc[i] + (size_t)kt * 256), (lptr_t*)(base + aoff[i]), 16, 0, 0);
    }
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  const unsigned g = lane >> 4, r = lane & 15u;
  if (nkt > 0) stage(0, 0);
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    wait_dma_and_barrier<0>();  // stage kt has landed for every wave; every wave has left the buffer about to be refilled
    if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
    const char* As = smem + cur * STAGE;
    h8 af[FM][NP];   // 2:4: the kept values' pieces;  DENSE: the pieces of k = 8 g .. 8 g + 7 of the first 32-k block ...
    h8 ag[FM][NP];   // ... and of the second (DENSE only)
    int idx[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const unsigned row = wave * TM + i * 16 + r;
      u4 v[4];
      if constexpr (DENSE) {
        // chunks (of 4 floats) 2 g, 2 g + 1 and 8 + 2 g, 9 + 2 g of the row
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const u4*>(As + row * 256u + 16u * ((8u * (c >> 1) + 2u * g + (c & 1)) ^ (row & 15u)));
        uint32_t pk[2][NP][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t xa = v[2 * h + (e >> 1)][2 * (e & 1)], xb = v[2 * h + (e >> 1)][2 * (e & 1) + 1];
            pk[h][0][e] = pack_hi16(xa, xb);
            if constexpr (NP >= 2) {
              const float ra = trunc_residual(xa), rb = trunc_residual(xb);
              pk[h][1][e] = pack_hi16(as_u32(ra), as_u32(rb));
              if constexpr (NP >= 3) pk[h][2][e] = pack_hi16(as_u32(trunc_residual(as_u32(ra))), as_u32(trunc_residual(as_u32(rb))));
            }
          }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          af[i][pl] = __builtin_bit_cast(h8, u4{pk[0][pl][0], pk[0][pl][1], pk[0][pl][2], pk[0][pl][3]});
          ag[i][pl] = __builtin_bit_cast(h8, u4{pk[1][pl][0], pk[1][pl][1], pk[1][pl][2], pk[1][pl][3]});
        }
        idx[i] = 0;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const u4*>(As + row * 256u + 16u * ((4u * g + c) ^ (row & 15u)));
        dense16_f32_to_operands<NP>(v, af[i], idx[i]);
      }
    }
    // B sweep: fragment j + 1's reads (4 per plane) in flight while fragment j's products run
    const unsigned bs_addr = (unsigned)(uintptr_t)(lds_char*)(As + SA);
    s4 t[2][NP][4];
    auto issue = [&](int j, int buf) {
      const unsigned q = r >> 2, pp = r & 3u;
      const unsigned a = bs_addr + b_off<64>(8u * g + q, (unsigned)j * 16u + 4u * pp);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                     "ds_read_b64_tr_b16 %2, %4 offset:4096\n\tds_read_b64_tr_b16 %3, %4 offset:4608"
                     : "=&v"(t[buf][pl][0]), "=&v"(t[buf][pl][1]), "=&v"(t[buf][pl][2]), "=&v"(t[buf][pl][3])
                     : "v"(a + (unsigned)pl * (unsigned)SBP)
                     : "memory");
    };
    issue(0, 0);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int c = j & 1;
      if (j + 1 < FN) {
        issue(j + 1, c ^ 1);
        if constexpr (NP == 3)
          asm volatile("s_waitcnt lgkmcnt(12)"
                       : "+v"(t[c][0][0]), "+v"(t[c][0][1]), "+v"(t[c][0][2]), "+v"(t[c][0][3]), "+v"(t[c][1][0]), "+v"(t[c][1][1]), "+v"(t[c][1][2]),
                         "+v"(t[c][1][3]), "+v"(t[c][2][0]), "+v"(t[c][2][1]), "+v"(t[c][2][2]), "+v"(t[c][2][3])
                       :: "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(8)"
                       : "+v"(t[c][0][0]), "+v"(t[c][0][1]), "+v"(t[c][0][2]), "+v"(t[c][0][3]), "+v"(t[c][1][0]), "+v"(t[c][1][1]), "+v"(t[c][1][2]),
                         "+v"(t[c][1][3])
                       :: "memory");
      } else {
        if constexpr (NP == 3)
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(t[c][0][0]), "+v"(t[c][0][1]), "+v"(t[c][0][2]), "+v"(t[c][0][3]), "+v"(t[c][1][0]), "+v"(t[c][1][1]), "+v"(t[c][1][2]),
                         "+v"(t[c][1][3]), "+v"(t[c][2][0]), "+v"(t[c][2][1]), "+v"(t[c][2][2]), "+v"(t[c][2][3])
                       :: "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(t[c][0][0]), "+v"(t[c][0][1]), "+v"(t[c][0][2]), "+v"(t[c][0][3]), "+v"(t[c][1][0]), "+v"(t[c][1][1]), "+v"(t[c][1][2]),
                         "+v"(t[c][1][3])
                       :: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      typedef short s16 __attribute__((ext_vector_type(16)));
      h16 bf[NP];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const s16 all = {t[c][pl][0][0], t[c][pl][0][1], t[c][pl][0][2], t[c][pl][0][3], t[c][pl][1][0], t[c][pl][1][1], t[c][pl][1][2], t[c][pl][1][3],
                         t[c][pl][2][0], t[c][pl][2][1], t[c][pl][2][2], t[c][pl][2][3], t[c][pl][3][0], t[c][pl][3][1], t[c][pl][3][2], t[c][pl][3][3]};
        bf[pl] = __builtin_bit_cast(h16, all);
      }
      // smallest terms first; consecutive instructions alternate between the FM accumulators (no back-to-back dependence)
      auto prod = [&](int pa, int pb) {
        if constexpr (DENSE) {
          typedef short s8v __attribute__((ext_vector_type(8)));
          const s8v lo = {t[c][pb][0][0], t[c][pb][0][1], t[c][pb][0][2], t[c][pb][0][3], t[c][pb][1][0], t[c][pb][1][1], t[c][pb][1][2], t[c][pb][1][3]};
          const s8v hi = {t[c][pb][2][0], t[c][pb][2][1], t[c][pb][2][2], t[c][pb][2][3], t[c][pb][3][0], t[c][pb][3][1], t[c][pb][3][2], t[c][pb][3][3]};
          const h8 b0 = __builtin_bit_cast(h8, lo), b1 = __builtin_bit_cast(h8, hi);
#pragma unroll
          for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<true>(af[i][pa], b0, acc[i][j]);
#pragma unroll
          for (int i = 0; i < FM; ++i) acc[i][j] = mfma16<true>(ag[i][pa], b1, acc[i][j]);
        } else {
#pragma unroll
          for (int i = 0; i < FM; ++i) acc[i][j] = smfmac16<true>(af[i][pa], bf[pb], acc[i][j], idx[i]);
        }
      };
      if constexpr (NP == 3) {
        prod(2, 0);
        prod(0, 2);
        prod(1, 1);
      }
      prod(1, 0);
      prod(0, 1);
      prod(0, 0);
    }
    cur ^= 1;
  }
  __syncthreads();
  // epilogue: the SMFMAC result map leaves 4 consecutive ROWS of one column per lane; the tile goes through LDS (pitch BN * 4 + 16
  // bytes, aliasing the ring) and out as 16-byte row pieces, alpha * acc + beta * C rounded once.
  constexpr unsigned CP = BN * 4 + 16;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float*>(smem + (wave * TM + i * 16 + 4u * g + q) * CP + (j * 16 + r) * 4u) = acc[i][j][q];
  __syncthreads();
  constexpr unsigned PPR = BN / 4;  // 16-byte pieces per row
  for (unsigned it = tid; it < (unsigned)BM * PPR; it += 64u * NW) {
    const unsigned row = it / PPR, pc = it - row * PPR;
    const int gr = m0 + (int)row, gc = n0 + (int)(4u * pc);
    if (gr > mlast || gc + 4 > p.N) continue;
    f4 v = *reinterpret_cast<const f4*>(smem + row * CP + pc * 16u);
    float* dst = C + (size_t)gr * p.N + gc;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] *= p.alpha;
    if (p.beta != 0.0f) {
      const f4 old = *reinterpret_cast<const f4*>(dst);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += p.beta * old[q];
    }
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
  }
}

// ---- pieces shared by the column-loop and span forms (one 16-row fragment per wave: FM = 1)
// the lane's 16 fp32 of its row -> operands.  2:4: v = 16 consecutive k (selection + pieces of the kept values, af / idx);
// DENSE: v[0..1] = k 8 g .. 8 g + 7 of the stage's first 32-k block, v[2..3] = of the second (pieces of all, af / ag).
template <int NP, bool DENSE>
__device__ __forceinline__ void split_a_operands(const u4 (&v)[4], h8 (&af)[NP], h8 (&ag)[NP], int& idx) {
  if constexpr (DENSE) {
    uint32_t pk[2][NP][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t xa = v[2 * h + (e >> 1)][2 * (e & 1)], xb = v[2 * h + (e >> 1)][2 * (e & 1) + 1];
        pk[h][0][e] = pack_hi16(xa, xb);
        if constexpr (NP >= 2) {
          const float ra = trunc_residual(xa), rb = trunc_residual(xb);
          pk[h][1][e] = pack_hi16(as_u32(ra), as_u32(rb));
          if constexpr (NP >= 3) pk[h][2][e] = pack_hi16(as_u32(trunc_residual(as_u32(ra))), as_u32(trunc_residual(as_u32(rb))));
        }
      }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      af[pl] = __builtin_bit_cast(h8, u4{pk[0][pl][0], pk[0][pl][1], pk[0][pl][2], pk[0][pl][3]});
      ag[pl] = __builtin_bit_cast(h8, u4{pk[1][pl][0], pk[1][pl][1], pk[1][pl][2], pk[1][pl][3]});
    }
    idx = 0;
  } else {
    dense16_f32_to_operands<NP>(v, af, idx);
  }
}

// B sweep of one 64-k (sub-)stage over FN 16-column fragments: the planes' images start at LDS address bs_addr, `pstride` bytes
// apart; fragment j + 1's reads (4 per plane) are in flight while fragment j's products run.
template <int FN, int NP, bool DENSE>
__device__ __forceinline__ void split_sweep1(unsigned bs_addr, unsigned pstride, unsigned lane, const h8 (&af)[NP], const h8 (&ag)[NP], int idx, f4 (&acc)[FN]) {
  const unsigned g = lane >> 4, r = lane & 15u;
  s4 t[2][NP][4];
  auto issue = [&](int j, int buf) {
    const unsigned q = r >> 2, pp = r & 3u;
    const unsigned a = bs_addr + b_off<64>(8u * g + q, (unsigned)j * 16u + 4u * pp);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
      asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                   "ds_read_b64_tr_b16 %2, %4 offset:4096\n\tds_read_b64_tr_b16 %3, %4 offset:4608"
                   : "=&v"(t[buf][pl][0]), "=&v"(t[buf][pl][1]), "=&v"(t[buf][pl][2]), "=&v"(t[buf][pl][3])
                   : "v"(a + (unsigned)pl * pstride)
                   : "memory");
  };
  issue(0, 0);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c = j & 1;
    if (j + 1 < FN) issue(j + 1, c ^ 1);
    if constexpr (NP == 3) {
      if (j + 1 < FN)
        asm volatile("s_waitcnt lgkmcnt(12)"
                     : "+v"(t[c][0][0]), "+v"(t[c][0][1]), "+v"(t[c][0][2]), "+v"(t[c][0][3]), "+v"(t[c][1][0]), "+v"(t[c][1][1]), "+v"(t[c][1][2]),
                       "+v"(t[c][1][3]), "+v"(t[c][2][0]), "+v"(t[c][2][1]), "+v"(t[c][2][2]), "+v"(t[c][2][3])
                     :: "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(t[c][0][0]), "+v"(t[c][0][1]), "+v"(t[c][0][2]), "+v"(t[c][0][3]), "+v"(t[c][1][0]), "+v"(t[c][1][1]), "+v"(t[c][1][2]),
                       "+v"(t[c][1][3]), "+v"(t[c][2][0]), "+v"(t[c][2][1]), "+v"(t[c][2][2]), "+v"(t[c][2][3])
                     :: "memory");
    } else {
      if (j + 1 < FN)
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(t[c][0][0]), "+v"(t[c][0][1]), "+v"(t[c][0][2]), "+v"(t[c][0][3]), "+v"(t[c][1][0]), "+v"(t[c][1][1]), "+v"(t[c][1][2]),
                       "+v"(t[c][1][3])
                     :: "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(t[c][0][0]), "+v"(t[c][0][1]), "+v"(t[c][0][2]), "+v"(t[c][0][3]), "+v"(t[c][1][0]), "+v"(t[c][1][1]), "+v"(t[c][1][2]),
                       "+v"(t[c][1][3])
                     :: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    auto prod = [&](int pa, int pb) {
      typedef short s8v __attribute__((ext_vector_type(8)));
      const s8v lo = {t[c][pb][0][0], t[c][pb][0][1], t[c][pb][0][2], t[c][pb][0][3], t[c][pb][1][0], t[c][pb][1][1], t[c][pb][1][2], t[c][pb][1][3]};
      const s8v hi = {t[c][pb][2][0], t[c][pb][2][1], t[c][pb][2][2], t[c][pb][2][3], t[c][pb][3][0], t[c][pb][3][1], t[c][pb][3][2], t[c][pb][3][3]};
      if constexpr (DENSE) {
        acc[j] = mfma16<true>(af[pa], __builtin_bit_cast(h8, lo), acc[j]);
        acc[j] = mfma16<true>(ag[pa], __builtin_bit_cast(h8, hi), acc[j]);
      } else {
        typedef short s16 __attribute__((ext_vector_type(16)));
        const s16 all = {lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7], hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
        acc[j] = smfmac16<true>(af[pa], __builtin_bit_cast(h16, all), acc[j], idx);
      }
    };
    if constexpr (NP == 3) {
      prod(2, 0);
      prod(0, 2);
      prod(1, 1);
    }
    prod(1, 0);
    prod(0, 1);
    prod(0, 0);
  }
}

// the epilogue of a 128 x BN fp32 tile held as 16-row fragments by eight waves: through LDS (pitch BN * 4 + 16 bytes) and out as
// 16-byte row pieces, alpha * acc + beta * C rounded once; columns n0 .. n0 + BN - 1, rows m0 .. (called between two barriers)
template <int BN, int FN>
__device__ __forceinline__ void split_store_tile(char* smem, float* C, const f4 (&acc)[FN], unsigned wave, unsigned lane, unsigned tid, int m0, int n0,
                                                 int mlast, int N, float alpha, float beta) {
  constexpr unsigned CP = BN * 4 + 16, PPR = BN / 4;
  const unsigned g = lane >> 4, r = lane & 15u;
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float*>(smem + (wave * 16u + 4u * g + q) * CP + (j * 16 + r) * 4u) = acc[j][q];
  __syncthreads();
  for (unsigned it = tid; it < 128u * PPR; it += 512u) {
    const unsigned row = it / PPR, pc = it - row * PPR;
    const int gr = m0 + (int)row, gc = n0 + (int)(4u * pc);
    if (gr > mlast || gc + 4 > N) continue;
    f4 v = *reinterpret_cast<const f4*>(smem + row * CP + pc * 16u);
    float* dst = C + (size_t)gr * N + gc;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] *= alpha;
    if (beta != 0.0f) {
      const f4 old = *reinterpret_cast<const f4*>(dst);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += beta * old[q];
    }
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
  }
}

// ---------------------------------------------------------------------------------------------
// n > 128: the COLUMN-LOOP form.  One workgroup owns 128 rows and ALL CT * 128 columns: a 64-k stage of A is selected and split
// ONCE (the operands stay in registers), then CT sub-stages follow, each with its own 128-column slice of B's planes -- instead of
// CT workgroups that each re-read the stage (from L2), select it again and split it again.  Two rings: A (2 x 32 KiB, refilled
// once per stage) and B (2 x planes x 16 KiB, refilled every sub-stage).  Issue order per sub-stage: B(s + 1), then -- at a stage's
// first sub-stage -- A(kt + 1); vmcnt retires in order, so at the top of sub-stage s everything up to B(s) has landed once only the
// A pieces issued after it remain (ct == 1), and a stage's first sub-stage waits for all (its A was issued CT sub-stages ago).
// Eight waves x 16 rows (FM = 1), accumulators for all CT * 8 column fragments in registers.
// ---------------------------------------------------------------------------------------------
template <int CT, int NP, bool DENSE>
__global__ __launch_bounds__(512) void spmma_f32_split_cols_kernel(const SplitArgs p) {
  constexpr int BM = 128, BN = 128, NW = 8, TM = 16, FN = BN / 16;
  constexpr int SA = BM * 256, SBP = 64 * BN * 2, SB = NP * SBP;
  constexpr int BRING = 2 * SA;
  constexpr int A_N = BM / 4, B_N = NP * (BN / 8);
  static_assert(A_N % NW == 0 && B_N % NW == 0, "equal DMA share per wave");
  constexpr int SLA = A_N / NW, SLB = B_N / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = lid / (unsigned)p.tiles_m, tile_m = lid - b * (unsigned)p.tiles_m;
  const int m0 = (int)tile_m * BM;
  const int nkt = p.K / 64;
  const float* A = p.A + (size_t)b * p.sA;
  const unsigned short* Bp = p.Bp + (size_t)b * p.sBp;
  float* C = p.C + (size_t)b * p.sC;
  const int mlast = p.Mrows - 1;
  const unsigned g = lane >> 4, r = lane & 15u;

  // per-lane DMA addresses as 32-bit offsets from two uniform bases (address state is what the 128 accumulators leave little room for)
  const char* const abase = reinterpret_cast<const char*>(A + (size_t)m0 * p.lda);
  const char* const bbase = reinterpret_cast<const char*>(Bp);
  unsigned asrc[SLA], bsrc[SLB], bpk[SLB];  // bpk: LDS offset in a B stage (bits 31:8, a multiple of 1 KiB) | first column of the lane's piece (bits 7:0)
#pragma unroll
  for (int i = 0; i < SLA; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    const unsigned row = 4u * t + (lane >> 4), cs = (lane & 15u) ^ (row & 15u);
    int lr = (int)row;
    lr = m0 + lr < mlast ? lr : mlast - m0;
    asrc[i] = (unsigned)lr * (unsigned)p.lda * 4u + 16u * cs;
  }
#pragma unroll
  for (int i = 0; i < SLB; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    const unsigned pl = t / (unsigned)(BN / 8), j = t - pl * (unsigned)(BN / 8);
    const unsigned panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
    const unsigned cs = (lane & 7u) ^ b_swz(kr);
    bsrc[i] = (unsigned)(((size_t)pl * p.plane + (size_t)kr * p.N) * 2);
    bpk[i] = (pl * SBP + panel * 8192u + (j & 7u) * 1024u) | (64u * panel + 8u * cs);
  }
  const size_t bstep = (size_t)64 * p.N * 2;
  auto stage_a = [&](int kt, int buf) {  // read once by the whole grid: non-temporal
#pragma unroll
    for (int i = 0; i < SLA; ++i) {
      const unsigned t = wave + (unsigned)NW * i;
      __builtin_amdgcn_global_load_lds((gptr_t*)(abase + asrc[i] + (size_t)kt * 256), (lptr_t*)(smem + buf * SA + t * 1024u), 16, 0, 2);
    }
  };
  auto stage_b = [&](int kt, int ct, int buf) {
#pragma unroll
    for (int i = 0; i < SLB; ++i) {
      int gc = ct * BN + (int)(bpk[i] & 0xffu);
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      __builtin_amdgcn_global_load_lds((gptr_t*)(bbase + bsrc[i] + (size_t)kt * bstep + (size_t)gc * 2), (lptr_t*)(smem + BRING + buf * SB + (bpk[i] & ~0xffu)), 16, 0, 0);
    }
  };

  f4 acc[CT][FN];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[c][j] = f4{0.f, 0.f, 0.f, 0.f};

  if (nkt > 0) {  // sub-stage -1's issue: B(0, 0), then A(0)
    stage_b(0, 0, 0);
    stage_a(0, 0);
  }
  h8 af[NP], ag[NP];
  int idx = 0;
  int bbuf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      __builtin_amdgcn_sched_barrier(0);  // sub-stages stay apart (the unrolled loop otherwise overlaps them and spills)
      // B(kt, 1) has landed once only the SLA pieces of A(kt + 1), issued after it, remain -- but the LAST stage issues no A(kt + 1):
      // there the youngest pieces in flight are B(kt, 1)'s own and the wait must be for all of them
      if (ct == 1 && kt + 1 < nkt) wait_dma_and_barrier<SLA>();
      else wait_dma_and_barrier<0>();
      // next sub-stage's B into the slot sub-stage s - 1 read (every wave has left it), then -- once per stage -- the next A
      if (ct + 1 < CT) stage_b(kt, ct + 1, bbuf ^ 1);
      else if (kt + 1 < nkt) stage_b(kt + 1, 0, bbuf ^ 1);
      if (ct == 0) {
        if (kt + 1 < nkt) stage_a(kt + 1, (kt + 1) & 1);
        // ---- this stage's A operands, once: selection (2:4) or plain split (DENSE), kept in registers for the CT sub-stages
        const char* As = smem + (kt & 1) * SA;
        const unsigned row = wave * TM + r;
        u4 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned chunk = DENSE ? 8u * (c >> 1) + 2u * g + (c & 1) : 4u * g + c;
          v[c] = *reinterpret_cast<const u4*>(As + row * 256u + 16u * (chunk ^ (row & 15u)));
        }
        split_a_operands<NP, DENSE>(v, af, ag, idx);
      }
      // ---- B sweep of this sub-stage's 128 columns
      split_sweep1<FN, NP, DENSE>((unsigned)(uintptr_t)(lds_char*)(smem + BRING + bbuf * SB), (unsigned)SBP, lane, af, ag, idx, acc[ct]);
      bbuf ^= 1;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // epilogue, one 128-column tile at a time through LDS (aliasing the rings)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    __syncthreads();
    split_store_tile<BN, FN>(smem, C, acc[ct], wave, lane, tid, m0, ct * BN, mlast, p.N, p.alpha, p.beta);
  }
}

template <int CT, int NP, bool DENSE>
static int launch_split_cols(const SplitArgs& a0, hipStream_t st) {
  SplitArgs a = a0;
  a.tiles_m = (a.Mrows + 127) / 128;
  a.tiles_n = 1;
  const size_t nwg = (size_t)a.tiles_m * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_spmma_fused_f32_split: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds = 2 * (size_t)128 * 256 + 2 * (size_t)NP * 64 * 128 * 2;
  static_assert(lds <= 160 * 1024 && (size_t)128 * (128 * 4 + 16) <= lds, "LDS budget");
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f32_split_cols_kernel<CT, NP, DENSE>), lds, "spmma_f32_split_cols_kernel")) return rc;
  spmma_f32_split_cols_kernel<CT, NP, DENSE><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(a);
  return check_launch("spmma_f32_split_cols_kernel");
}

// ---------------------------------------------------------------------------------------------
// k % 64 != 0 (the 7 x 7 x 3 stem layer, k = 147: rows of 588 bytes): the SPAN form, as the fp16 fused kernels have it.  A is one
// tall contiguous matrix (lda == k), so a tile's 128 rows are ONE contiguous span of 128 * k * 4 bytes that starts on a 512-byte
// boundary and reaches LDS by plain 1 KiB LDS-DMA pieces whatever the row pitch; B's planes arrive whole with it (rows at or beyond
// k from a zero page: a 0 x inf must not poison finite outputs); one wait, one barrier, then every stage is computed out of LDS:
// the lane reads its 16 fp32 with 4-byte LDS reads at row * k * 4 + ..., zeroes what lies at or beyond k (the virtual zeros that
// complete a ragged strip, oracle: strip_select) and goes on as the other forms do.  n <= 128; span + planes inside the LDS.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) const unsigned char sm_split_zero_page[256] = {0};

template <int BN, int NP, bool DENSE>
__global__ __launch_bounds__(512) void spmma_f32_split_span_kernel(const SplitArgs p, const unsigned span_lds /*bytes reserved for the A span*/,
                                                                   const size_t a_bytes /*bytes of A*/) {
  constexpr int BM = 128, NW = 8, FN = BN / 16;
  constexpr int SBP = 64 * BN * 2;  // one plane of one stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, p.tiles_n == 1 && !p.xcd_ranges);  // (mma_tile.h)
  const unsigned tile_m = lid / (unsigned)p.tiles_n, tile_n = lid - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = (p.K + 63) / 64;
  const char* A = reinterpret_cast<const char*>(p.A);
  const unsigned rowbytes = (unsigned)p.K * 4u;
  const int rows = p.Mrows - m0 < BM ? p.Mrows - m0 : BM;
  const int mlast = p.Mrows - 1;
  const unsigned g = lane >> 4, r = lane & 15u;
  {  // ---- A span
    const size_t s0 = (size_t)m0 * rowbytes;
    const unsigned len = (unsigned)rows * rowbytes;
    const unsigned np = (len + 1023u) / 1024u;
    const size_t last16 = a_bytes - 16;
    for (unsigned pc = wave; pc < np; pc += NW) {
      size_t off = s0 + (size_t)pc * 1024u + 16u * lane;
      off = off < last16 ? off : last16;
      __builtin_amdgcn_global_load_lds((gptr_t*)(A + off), (lptr_t*)(smem + pc * 1024u), 16, 0, 2);
    }
  }
  char* const Bimg = smem + span_lds;  // [stage][plane][64 x BN image]
  {  // ---- B planes, whole
    constexpr int B_N = BN / 8;
    const int npb = nkt * NP * B_N;
    for (int t = (int)wave; t < npb; t += NW) {
      const int kt = t / (NP * B_N), rem = t - kt * (NP * B_N), pl = rem / B_N, j = rem - pl * B_N;
      const unsigned panel = (unsigned)j >> 3, kr = 8u * ((unsigned)j & 7u) + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ b_swz(kr);
      int gc = n0 + (int)(64u * panel + 8u * cs);
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      const int krow = kt * 64 + (int)kr;
      const char* src = krow < p.K ? reinterpret_cast<const char*>(p.Bp + (size_t)pl * p.plane + (size_t)krow * p.N + gc)
                                   : reinterpret_cast<const char*>(sm_split_zero_page) + 16u * (lane & 7u);
      __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(Bimg + (kt * NP + pl) * SBP + panel * 8192u + ((unsigned)j & 7u) * 1024u), 16, 0, 0);
    }
  }
  wait_dma_and_barrier<0>();

  f4 acc[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
  int row = (int)(wave * 16u + r);
  row = row < rows ? row : rows - 1;  // rows past the edge re-read the last valid one (their outputs are never stored)
  const char* const arow = smem + (unsigned)row * rowbytes;
  for (int kt = 0; kt < nkt; ++kt) {
    u4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // 2:4: k = 64 kt + 16 g + 4 c ..;  DENSE: c < 2: k = 64 kt + 8 g + 4 c ..,  c >= 2: k = 64 kt + 32 + 8 g + 4 (c - 2) ..
      const int k0 = kt * 64 + (DENSE ? 32 * (c >> 1) + 8 * (int)g + 4 * (c & 1) : 16 * (int)g + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // (reads at or beyond k are masked; they stay inside the LDS allocation: span_lds covers 128 rows + 256 bytes)
        const uint32_t x = *reinterpret_cast<const uint32_t*>(arow + 4u * (unsigned)(k0 + e));
        v[c][e] = k0 + e < p.K ? x : 0u;
      }
    }
    h8 af[NP], ag[NP];
    int idx;
    split_a_operands<NP, DENSE>(v, af, ag, idx);
    split_sweep1<FN, NP, DENSE>((unsigned)(uintptr_t)(lds_char*)(Bimg + kt * NP * SBP), (unsigned)SBP, lane, af, ag, idx, acc);
  }
  __syncthreads();
  split_store_tile<BN, FN>(smem, p.C, acc, wave, lane, tid, m0, n0, mlast, p.N, p.alpha, p.beta);
}

template <int BN, int NP, bool DENSE>
static int launch_split_span(const SplitArgs& a0, hipStream_t st) {
  SplitArgs a = a0;
  a.tiles_m = (a.Mrows + 127) / 128;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  const size_t kc = ((size_t)a.K + 63) / 64 * 64;
  const size_t span_lds = ((size_t)128 * a.K * 4 + 256 + 1023) / 1024 * 1024;  // whole pieces; >= 256 bytes past the last row
  const size_t lds_main = span_lds + (size_t)NP * kc * BN * 2;
  constexpr size_t lds_epi = (size_t)128 * (BN * 4 + 16);
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  if (lds > 160 * 1024 || nwg > 0x7fffffffu) {
    set_error("sm_spmma_fused_f32_split: k too long for the span form (use sm_spmma_fused_f32)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f32_split_span_kernel<BN, NP, DENSE>), 160 * 1024, "spmma_f32_split_span_kernel")) return rc;
  spmma_f32_split_span_kernel<BN, NP, DENSE><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(a, (unsigned)span_lds, (size_t)a.Mrows * a.K * 4);
  return check_launch("spmma_f32_split_span_kernel");
}

// B (fp32, [k][n] row-major per batch) -> planes of truncated bfloat16 pieces, 8 elements per thread
template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ B, unsigned short* __restrict__ P, size_t items /* of 8 */, size_t plane) {
  for (size_t it = (size_t)blockIdx.x * 256u + threadIdx.x; it < items; it += (size_t)gridDim.x * 256u) {
    const u4 lo = *reinterpret_cast<const u4*>(B + it * 8), hi = *reinterpret_cast<const u4*>(B + it * 8 + 4);
    const uint32_t x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    uint32_t o[NP][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t a = x[2 * e], b = x[2 * e + 1];
      o[0][e] = pack_hi16(a, b);
      if constexpr (NP >= 2) {
        const float ra = trunc_residual(a), rb = trunc_residual(b);
        o[1][e] = pack_hi16(as_u32(ra), as_u32(rb));
        if constexpr (NP >= 3) {
          o[2][e] = pack_hi16(as_u32(trunc_residual(as_u32(ra))), as_u32(trunc_residual(as_u32(rb))));
        }
      }
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u4*>(P + (size_t)pl * plane + it * 8) = u4{o[pl][0], o[pl][1], o[pl][2], o[pl][3]};
  }
}

template <int BN, int NP, int NW, bool DENSE = false>
static int launch_split(const SplitArgs& a0, hipStream_t st) {
  SplitArgs a = a0;
  a.tiles_m = (a.Mrows + 127) / 128;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_spmma_fused_f32_split: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t stage_bytes = (size_t)128 * 256 + (size_t)NP * 64 * BN * 2;
  constexpr size_t lds_epi = (size_t)128 * (BN * 4 + 16);
  const size_t lds_main = (a.K / 64 < 2 ? 1 : 2) * stage_bytes;
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  constexpr size_t lds_max = 2 * stage_bytes > lds_epi ? 2 * stage_bytes : lds_epi;
  static_assert(lds_max <= 160 * 1024, "LDS budget");
  static LdsOptIn lds_optin, lds_optin_nt;
  if (a.tiles_n == 1) {  // A is read once by the whole grid: non-temporal
    if (const int rc = ensure_dyn_lds(lds_optin_nt, reinterpret_cast<const void*>(&spmma_f32_split_kernel<BN, NP, true, NW, DENSE>), lds_max, "spmma_f32_split_kernel")) return rc;
    spmma_f32_split_kernel<BN, NP, true, NW, DENSE><<<dim3((unsigned)nwg), dim3(64 * NW), lds, st>>>(a);
  } else {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f32_split_kernel<BN, NP, false, NW, DENSE>), lds_max, "spmma_f32_split_kernel")) return rc;
    spmma_f32_split_kernel<BN, NP, false, NW, DENSE><<<dim3((unsigned)nwg), dim3(64 * NW), lds, st>>>(a);
  }
  return check_launch("spmma_f32_split_kernel");
}

}  // namespace sm

extern "C" int sm_spmma_fused_f32_split_workspace(size_t n, size_t k, size_t batch, size_t strideB, int planes, size_t* bytes) {
  if (!bytes || (planes != 2 && planes != 3)) {
    sm::set_error("sm_spmma_fused_f32_split_workspace: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  size_t e = 0, t = 0;
  if (__builtin_mul_overflow(n, k, &e) || __builtin_mul_overflow(e, strideB ? batch : (size_t)1, &e) || __builtin_mul_overflow(e, (size_t)planes * 2, &t)) {
    sm::set_error("sm_spmma_fused_f32_split_workspace: size overflows");
    return SM_STATUS_NOT_SUPPORTED;
  }
  *bytes = t;
  return SM_STATUS_SUCCESS;
}

// prepared: false = split B into `workspace` first (one streaming pass per call), then multiply; true = `workspace` already holds B's planes
// (sm_spmma_fused_f32_split_prepare: weights that stay the same across calls are split once) and B is not read at all (it may be null)
static int f32_split_product(bool dense, const float* A, const float* B, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                             size_t strideB, size_t strideC, int planes, void* workspace, size_t workspace_bytes, float alpha, float beta,
                             sm_stream_t stream, bool prepared = false) {
  using namespace sm;
  if (!A || (!B && !prepared) || !C || lda < k || (planes != 2 && planes != 3)) {
    set_error("sm_spmma_fused_f32_split: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  // ragged k: the span form -- one tall contiguous A (lda == k, shared B, batches back to back), n <= 128, span + planes inside the LDS
  const bool span = (k % 64 != 0 || lda % 4 != 0) && k != 0 && lda == k && n <= 128 && (batch == 1 || (strideB == 0 && strideA == m * lda && strideC == m * n)) &&
                    (m * batch * k * 4) % 16 == 0 &&
                    ((size_t)128 * k * 4 + 256 + 1023) / 1024 * 1024 + (size_t)planes * ((k + 63) / 64 * 64) * (n <= 64 ? 64 : 128) * 2 <= 160 * 1024;
  if (k == 0 || (!span && (k % 64 != 0 || lda % 4 != 0)) || n % 8 != 0 || strideA % 4 != 0 || strideB % 8 != 0 || strideC % 4 != 0 || !aligned16(A) || (!prepared && !aligned16(B)) ||
      !aligned16(C) || m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || lda > 0x7fffffffull) {
    set_error("sm_spmma_fused_f32_split: needs k %% 64 == 0 (or a ragged k with n <= 128, lda == k and one tall A that fits the span form), n %% 8 == 0 and "
              "16-byte aligned A, B and C (use sm_spmma_fused_f32)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  size_t need = 0;
  if (const int rc = sm_spmma_fused_f32_split_workspace(n, k, batch, strideB, planes, &need)) return rc;
  if (!workspace || workspace_bytes < need || !aligned16(workspace)) {
    set_error("sm_spmma_fused_f32_split: workspace of %zu bytes (16-byte aligned) required", need);
    return SM_STATUS_INVALID_VALUE;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t nb = strideB ? batch : 1;
  if (nb > 1 && strideB != k * n) {
    set_error("sm_spmma_fused_f32_split: a strided B must be packed (strideB == k * n)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  const size_t plane = nb * k * n, items = plane / 8;
  unsigned short* P = (unsigned short*)workspace;
  const unsigned grid = (unsigned)std::min<size_t>((items + 255) / 256, 4096);
  if (!prepared) {
    if (planes == 3) split_planes_kernel<3><<<dim3(grid), dim3(256), 0, st>>>(B, P, items, plane);
    else split_planes_kernel<2><<<dim3(grid), dim3(256), 0, st>>>(B, P, items, plane);
    if (const int rc = check_launch("split_planes_kernel")) return rc;
  }
  SplitArgs a = {};
  a.xcd_ranges = tuning_int("SM_SPLIT_XCD", 0);
  a.A = A; a.Bp = P; a.C = C;
  a.sA = strideA; a.sBp = strideB ? k * n : 0; a.plane = plane; a.sC = strideC;
  a.Mrows = (int)m; a.N = (int)n; a.K = (int)k; a.lda = (int)lda; a.batch = (int)batch;
  a.alpha = alpha; a.beta = beta;
  if (batch > 1 && strideB == 0 && strideA == m * lda && strideC == m * n) {  // one tall matrix
    a.Mrows = (int)(m * batch);
    a.batch = 1;
  }
  if (span) {
    a.Mrows = (int)(m * batch);
    a.batch = 1;
    if (dense) {
      if (planes == 3) return n <= 64 ? launch_split_span<64, 3, true>(a, st) : launch_split_span<128, 3, true>(a, st);
      return n <= 64 ? launch_split_span<64, 2, true>(a, st) : launch_split_span<128, 2, true>(a, st);
    }
    if (planes == 3) return n <= 64 ? launch_split_span<64, 3, false>(a, st) : launch_split_span<128, 3, false>(a, st);
    return n <= 64 ? launch_split_span<64, 2, false>(a, st) : launch_split_span<128, 2, false>(a, st);
  }
  // 128 < n <= 256: the column-loop form (A selected / split once per stage for both 128-column halves).  Not beyond: four column
  // tiles need 128 accumulator registers per lane (the kernel spills) and leave one workgroup per 128 rows -- 49 workgroups for the
  // 196 x 512 layers -- where four column-tile workgroups per 128 rows at least fill the chip.
  // (its per-lane DMA sources are 32-bit offsets from two uniform bases: the last plane's last stage row and a tile's last A row must
  // stay below 4 GiB, or the call takes the 128-column tiles below, whose addresses are 64-bit)
  const bool cols_offsets_fit = (size_t)(planes - 1) * plane * 2 + (size_t)64 * n * 2 < ((size_t)1 << 32) && (size_t)128 * lda * 4 < ((size_t)1 << 32);
  if (n > 128 && n <= 256 && cols_offsets_fit && tuning_int("SM_F32_SPLIT_COLS", 1)) {
    if (dense) return planes == 3 ? launch_split_cols<2, 3, true>(a, st) : launch_split_cols<2, 2, true>(a, st);
    return planes == 3 ? launch_split_cols<2, 3, false>(a, st) : launch_split_cols<2, 2, false>(a, st);
  }
  if (dense) {  // every element multiplied: the dense comparator of the 2:4 form (sm_gemm_rowmajor_f32_split)
    if (planes == 3) return n <= 64 ? launch_split<64, 3, 8, true>(a, st) : launch_split<128, 3, 8, true>(a, st);
    return n <= 64 ? launch_split<64, 2, 8, true>(a, st) : launch_split<128, 2, 8, true>(a, st);
  }
#ifdef SM_TUNING
  if (tuning_int("SM_F32_SPLIT_NW", 8) == 4) {  // A/B: four waves of 32 rows (one per SIMD) instead of eight of 16
    if (planes == 3) return n <= 64 ? launch_split<64, 3, 4>(a, st) : launch_split<128, 3, 4>(a, st);
    return n <= 64 ? launch_split<64, 2, 4>(a, st) : launch_split<128, 2, 4>(a, st);
  }
#endif
  if (planes == 3) return n <= 64 ? launch_split<64, 3, 8>(a, st) : launch_split<128, 3, 8>(a, st);
  return n <= 64 ? launch_split<64, 2, 8>(a, st) : launch_split<128, 2, 8>(a, st);
}

extern "C" int sm_spmma_fused_f32_split(const float* A, const float* B, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                                        size_t strideB, size_t strideC, int planes, void* workspace, size_t workspace_bytes, float alpha, float beta,
                                        sm_stream_t stream) {
  return f32_split_product(false, A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, planes, workspace, workspace_bytes, alpha, beta, stream);
}
extern "C" int sm_gemm_rowmajor_f32_split(const float* A, const float* B, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                                          size_t strideB, size_t strideC, int planes, void* workspace, size_t workspace_bytes, float alpha, float beta,
                                          sm_stream_t stream) {
  return f32_split_product(true, A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, planes, workspace, workspace_bytes, alpha, beta, stream);
}

// B's bfloat16 planes once, for operands that stay the same across calls (weights): sm_spmma_fused_f32_split_prepare writes them into
// `workspace` (sm_spmma_fused_f32_split_workspace bytes), sm_spmma_fused_f32_split_prepared multiplies from them -- the same kernels, the
// same C bit for bit as sm_spmma_fused_f32_split, without the per-call streaming pass over B.
extern "C" int sm_spmma_fused_f32_split_prepare(const float* B, size_t n, size_t k, size_t batch, size_t strideB, int planes, void* workspace,
                                                size_t workspace_bytes, sm_stream_t stream) {
  using namespace sm;
  size_t need = 0;
  if (!B || (planes != 2 && planes != 3) || n % 8 != 0 || !aligned16(B)) {
    set_error("sm_spmma_fused_f32_split_prepare: invalid argument (n %% 8 == 0, a 16-byte aligned B, planes 2 or 3)");
    return SM_STATUS_INVALID_VALUE;
  }
  if (const int rc = sm_spmma_fused_f32_split_workspace(n, k, batch, strideB, planes, &need)) return rc;
  if (!workspace || workspace_bytes < need || !aligned16(workspace) || (strideB && strideB != k * n)) {
    set_error("sm_spmma_fused_f32_split_prepare: workspace of %zu bytes (16-byte aligned) required; a strided B must be packed", need);
    return SM_STATUS_INVALID_VALUE;
  }
  const size_t nb = strideB ? batch : 1, plane = nb * k * n, items = plane / 8;
  if (items == 0) return SM_STATUS_SUCCESS;
  const unsigned grid = (unsigned)std::min<size_t>((items + 255) / 256, 4096);
  if (planes == 3) split_planes_kernel<3><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(B, (unsigned short*)workspace, items, plane);
  else split_planes_kernel<2><<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(B, (unsigned short*)workspace, items, plane);
  return check_launch("split_planes_kernel");
}
extern "C" int sm_spmma_fused_f32_split_prepared(const float* A, const void* planes_workspace, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                                                 size_t strideA, size_t strideB, size_t strideC, int planes, size_t workspace_bytes, float alpha, float beta,
                                                 sm_stream_t stream) {
  return f32_split_product(false, A, nullptr, C, m, n, k, lda, batch, strideA, strideB, strideC, planes, const_cast<void*>(planes_workspace), workspace_bytes, alpha, beta,
                           stream, true);
}
