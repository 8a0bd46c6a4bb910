// spmma_f16.hip -- the 2:4 sparse x dense matmul on gfx950's native sparse matrix instruction
// v_smfmac_f32_16x16x64_f16 (fp32 accumulate, one rounding to fp16).  Replaces cusparseLtMatmul as
// include/sparsify.me/spmma.hxx:112-113 calls it: C[m x n] = alpha * A * B + beta * C, row-major,
// A the compressed blob of sm_compress24_f16 (stage-major values [kc/64][M][32] + metadata [kc/64][M][8 B]).
//
// Per 128 dense k of a row the kernel moves 128 B of kept values + 16 B of metadata instead of the
// dense kernel's 256 B, and issues half the matrix instructions: the hardware multiplies each kept
// value with the B row its 2-bit index selects, so nothing is expanded and no zero is multiplied.
// Tile: BM x BN outputs, BK = 128 dense k (two SMFMAC k-steps) per stage, 256 threads = 4 waves.
// Global -> registers -> swizzled LDS images (mma_tile.h) with the next stage's loads in flight
// during the current stage's SMFMACs.  The sparse operand must be srcA, so a lane ends with four
// consecutive ROWS of one column; the epilogue transposes through LDS and stores 16-byte row pieces.
#include <stdlib.h>

#include <vector>

#include "spmma_args.h"

#ifndef SM_NT_A
#define SM_NT_A 0
#endif

namespace sm {

// 256 zero bytes in device memory: the B rows of a K tail (k % 64 != 0: stage rows at or beyond k) are DMA'd from
// here, so the blob's zero padding meets zeros, never a clamped real row (0 * inf would poison finite outputs).
__device__ __attribute__((aligned(256))) const unsigned char sm_zero_page[256] = {0};


template <int BM, int BN, int WM, int WN, bool BF = false>
__global__ __launch_bounds__(256) void spmma_f16_kernel(const SpmmaArgs p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int A_CH = BM * 8 / 256;        // 16-byte value chunks per thread per stage
  constexpr int M_CH = (BM * 2 + 255) / 256;  // 8-byte metadata pieces per thread per stage
  constexpr int B_CH = 128 * (BN / 8) / 256;  // 128 k-rows x BN/8 chunks
  constexpr int CPITCH = BN * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                     // BM x 128 B
  char* Ms = smem + BM * 128;          // BM x 16 B
  char* Bs = smem + BM * 144;          // BN/64 panels x 128 rows x 128 B

  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned wm = wave / WN, wn = wave % WN;

  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned gb = lid / tiles, trem = lid - gb * tiles;
  const unsigned grp = gb / (unsigned)p.batch, b = gb - grp * (unsigned)p.batch;  // problem of the group, grid batch in it
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;

  const size_t row_base = (size_t)b * p.m;  // first blob row of this grid batch
  const char* vals = p.vals[grp] + row_base * 64;  // + stage * Mtot * 64 per 64-k stage (stage-major values)
  const char* meta = p.meta[grp] + row_base * 8;   // + stage * Mtot * 8 per 64-k stage
  const half_t* B = p.B[grp] + (size_t)b * p.sB;
  half_t* C = p.C[grp] + (size_t)b * p.sC;

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  u4 ra[A_CH], rb[B_CH];
  u2 rm[M_CH];
  const bool b_vec = (p.N % 8 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15u) == 0);

  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      // a 128-k stage of this kernel = two 64-k planes of the blob, 64 bytes per row in each
      const unsigned q = tid + 256u * i, row = q >> 3, ch = q & 7u;
      const int gr = m0 + (int)row, st64 = kt * 2 + (int)(ch >> 2);
      u4 v = {0u, 0u, 0u, 0u};
      if (gr < p.Mrows && st64 * 64 < p.kc)
        v = *reinterpret_cast<const u4*>(vals + ((size_t)st64 * p.Mtot + (size_t)gr) * 64 + 16u * (ch & 3u));
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < M_CH; ++i) {
      const unsigned q = tid + 256u * i, row = q >> 1, part = q & 1u;
      const int gr = m0 + (int)row, st64 = kt * 2 + (int)part;  // the 64-k stage this 8-byte piece belongs to
      u2 v = {0x44444444u, 0x44444444u};
      if (row < (unsigned)BM && gr < p.Mrows && st64 * 64 < p.kc)
        v = *reinterpret_cast<const u2*>(meta + ((size_t)st64 * p.Mtot + (size_t)gr) * 8);
      rm[i] = v;
    }
    const int k0 = kt * 128;
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const unsigned q = tid + 256u * i, kr = q / (BN / 8), cn = q % (BN / 8);
      const int gk = k0 + (int)kr, gc = n0 + 8 * (int)cn;
      u4 v = {0u, 0u, 0u, 0u};
      if (gk < p.K && gc < p.N) {
        const half_t* src = B + (size_t)gk * p.N + gc;
        if (b_vec && gc + 8 <= p.N) {
          v = *reinterpret_cast<const u4*>(src);
        } else {
          h8 e;
#pragma unroll
          for (int t = 0; t < 8; ++t) e[t] = (gc + t < p.N) ? src[t] : (half_t)0.0f;
          v = __builtin_bit_cast(u4, e);
        }
      }
      rb[i] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const unsigned q = tid + 256u * i, row = q >> 3, ch = q & 7u;
      *reinterpret_cast<u4*>(As + a_off(row, ch)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < M_CH; ++i) {
      const unsigned q = tid + 256u * i, row = q >> 1, part = q & 1u;
      if (row < (unsigned)BM) *reinterpret_cast<u2*>(Ms + row * 16u + 8u * part) = rm[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const unsigned q = tid + 256u * i, kr = q / (BN / 8), cn = q % (BN / 8);
      *reinterpret_cast<u4*>(Bs + b_off<128>(kr, 8u * cn)) = rb[i];
    }
  };

  const int nkt = (p.kc + 127) / 128;
  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < nkt) gload(kt + 1);
    const int nstep = (p.kc - kt * 128) >= 128 ? 2 : 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s < nstep) {
        h8 af[FM];
        int idx[FM];
        h16 bf[FN];
        const unsigned g = lane >> 4;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const unsigned row = wm * TM + i * 16 + (lane & 15u);
          af[i] = *reinterpret_cast<const h8*>(As + a_off(row, 4u * s + g));
          idx[i] = (int)*reinterpret_cast<const unsigned short*>(Ms + row * 16u + 8u * s + 2u * g);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const unsigned col0 = wn * TN + j * 16, kr0 = 64u * s + 8u * g;
          const s4 v0 = b_read_tr<128>(Bs, kr0, col0, lane);
          const s4 v1 = b_read_tr<128>(Bs, kr0 + 4u, col0, lane);
          const s4 v2 = b_read_tr<128>(Bs, kr0 + 32u, col0, lane);
          const s4 v3 = b_read_tr<128>(Bs, kr0 + 36u, col0, lane);
          typedef short s16 __attribute__((ext_vector_type(16)));
          const s16 all = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3],
                           v2[0], v2[1], v2[2], v2[3], v3[0], v3[1], v3[2], v3[3]};
          bf[j] = __builtin_bit_cast(h16, all);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = smfmac16<BF>(af[i], bf[j], acc[i][j], idx[i]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[rows 4*(lane>>4) + r][col lane&15] of each fragment
  const bool c_vec = (p.N % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
  if (p.beta == 0.0f && c_vec) {
    char* Cs = smem;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const unsigned row = wm * TM + i * 16 + 4u * (lane >> 4), col = wn * TN + j * 16 + (lane & 15u);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<half_t*>(Cs + (row + r) * CPITCH + col * 2) = to_elt<BF>(p.alpha * acc[i][j][r]);
      }
    __syncthreads();
    constexpr int C_CH = BM * (BN / 8) / 256;
#pragma unroll
    for (int i = 0; i < C_CH; ++i) {
      const unsigned q = tid + 256u * i, row = q / (BN / 8), cn = q % (BN / 8);
      const int gr = m0 + (int)row, gc = n0 + 8 * (int)cn;
      if (gr >= p.Mrows || gc >= p.N) continue;
      const u4 v = *reinterpret_cast<const u4*>(Cs + row * CPITCH + cn * 16);
      half_t* dst = C + (size_t)gr * p.N + gc;
      if (gc + 8 <= p.N) {
        __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst));
      } else {
        const h8 e = __builtin_bit_cast(h8, v);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (gc + t < p.N) dst[t] = e[t];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int gc = n0 + (int)(wn * TN + j * 16 + (lane & 15u));
        if (gc >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gr = m0 + (int)(wm * TM + i * 16 + 4u * (lane >> 4)) + r;
          if (gr >= p.Mrows) continue;
          half_t* dst = C + (size_t)gr * p.N + gc;
          float v = p.alpha * acc[i][j][r];
          if (p.beta != 0.0f) v += p.beta * to_f32<BF>(*dst);
          *dst = to_elt<BF>(v);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Fast path (K % 64 == 0, N % 8 == 0, 16-byte aligned B): LDS-DMA double-buffered pipeline.
// Stage = 64 dense k: A values [BM][64 B] + metadata [BM][8 B] + B [64][BN], all brought in by
// global_load_lds (no VGPR staging, no ds_write); two LDS buffers, the DMA of stage t+1 is in
// flight while stage t's SMFMACs run; one barrier per stage.  LDS images are lane-linear for the
// DMA, so the bank swizzles are applied to the per-lane SOURCE address and again on the read.
// Rows / columns past the matrix edge are clamped to the last valid one (their products land in
// outputs that are never stored), so no lane is ever predicated off.
// ---------------------------------------------------------------------------------------------


template <int BM, int BN, int WM, int WN, int NS, bool BF = false>
__global__ __launch_bounds__(64 * WM * WN) void spmma_f16_dma_kernel(const SpmmaArgs p) {
  constexpr int NW = WM * WN;  // waves per workgroup: 4 for big grids, 8 / 16 when few tiles exist
  static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(FM >= 1 && FN >= 1, "wave tile");
  constexpr int SA = BM * 64, SM_ = BM * 8, SB = 64 * BN * 2, STAGE = SA + SM_ + SB;
  // One stage = W DMA wave-instructions of 1 KiB: A_N (16 rows x 64 B each), M_N (128 rows x 8 B of
  // the stage-major metadata plane: contiguous), B_N (8 k-rows x 128 B).  Instruction t belongs to
  // wave t % NW (slot t / NW).
  static_assert(BM % 128 == 0, "metadata DMA moves 128 rows per instruction");
  constexpr int A_N = BM / 16, M_N = BM / 128, B_N = BN / 8, W = A_N + M_N + B_N;
  constexpr int SL = (W + NW - 1) / NW;  // slots per wave
  constexpr int LPS = W / NW;            // least DMA instructions any wave issues per stage (vmcnt unit)
  static_assert(LPS >= 1, "every wave must issue at least one DMA per stage");
  constexpr int NT_A = SM_NT_A;  // cache policy bits of the A-side DMA (2 = nt)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned wm = wave / WN, wn = wave % WN;
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned gb = lid / tiles, trem = lid - gb * tiles;
  const unsigned grp = gb / (unsigned)p.batch, b = gb - grp * (unsigned)p.batch;  // problem of the group, grid batch in it
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;

  const size_t row_base = (size_t)b * p.m;
  const char* vals = p.vals[grp] + row_base * 64;  // plane of stage 0; + Mtot * 64 per stage (stage-major values)
  const char* meta = p.meta[grp] + row_base * 8;  // plane of stage 0; + Mtot * 8 per stage
  const half_t* B = p.B[grp] + (size_t)b * p.sB;
  half_t* C = p.C[grp] + (size_t)b * p.sC;
  const int mlast = p.Mrows - 1;

  // per-slot source address of stage 0 (advanced by `step` bytes per stage) and LDS offset in a stage
  const char* src[SL];
  size_t step[SL];
  unsigned loff[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    if (t < (unsigned)A_N) {
      const unsigned row = 16u * t + (lane >> 2), cs = (lane & 3u) ^ a64_swz(row);
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      src[i] = vals + (size_t)gr * 64 + 16u * cs;
      step[i] = p.Mtot * 64;
      loff[i] = t * 1024u;
    } else if (t < (unsigned)(A_N + M_N)) {
      // lane moves the 16 bytes of tile rows 2*lane, 2*lane+1 (clamped to the batch's last row pair)
      const unsigned u = t - A_N;
      size_t off = ((size_t)m0 + 128u * u + 2u * lane) * 8;
      const size_t last = (size_t)p.Mrows * 8 - 16;
      off = off < last ? off : last;
      src[i] = meta + off;
      step[i] = p.Mtot * 8;
      loff[i] = SA + u * 1024u;
    } else {
      const unsigned j = t - (A_N + M_N), panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ b_swz(kr);
      int gc = n0 + (int)(64u * panel + 8u * cs);
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      src[i] = reinterpret_cast<const char*>(B + (size_t)kr * p.N + gc);
      step[i] = (size_t)64 * p.N * 2;
      loff[i] = SA + SM_ + panel * 8192u + (j & 7u) * 1024u;
    }
  }

  const bool ktail = (p.K & 63) != 0;
  const int nkt_last = p.kc / 64 - 1;
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const unsigned t = wave + (unsigned)NW * i;  // wave-uniform
      if (t >= (unsigned)W) continue;
      gptr_t* g = (gptr_t*)(src[i] + (size_t)kt * step[i]);
      lptr_t* l = (lptr_t*)(base + loff[i]);
      if (ktail && kt == nkt_last && t >= (unsigned)(A_N + M_N)) {  // B rows of the K tail: zeros
        const unsigned kr = 8u * ((t - (unsigned)(A_N + M_N)) & 7u) + (lane >> 3);
        if (kt * 64 + (int)kr >= p.K) g = (gptr_t*)(sm_zero_page + 16u * (lane & 7u));
      }
      // A values and metadata are read exactly once (non-temporal: keep them from evicting B, which
      // every workgroup re-reads from L2); B uses the default policy.
      if (t < (unsigned)(A_N + M_N))
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, NT_A);
      else
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    }
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  // Ring of NS stage buffers.  Iteration kt: (1) wait until this wave's DMA of stage kt has landed
  // (stages kt+1 .. kt+NS-2 may stay in flight: counted vmcnt), (2) barrier, (3) refill the buffer
  // read in iteration kt-1 with stage kt+NS-1, (4) compute stage kt.  One barrier per stage.
  const int nkt = p.kc / 64;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) stage(s, s);
  int cur = 0, fill = NS - 1;  // buffer of stage kt, buffer of stage kt+NS-1
  SM_T(unsigned long long tw = 0, ti = 0, tc = 0; unsigned long long st0 = sm_stamp(); const unsigned long long tstart = st0;)
  for (int kt = 0; kt < nkt; ++kt) {
    const int ahead = (nkt - 1 - kt) < (NS - 2) ? (nkt - 1 - kt) : (NS - 2);
    if (NS >= 4 && ahead == 2) wait_dma_and_barrier<2 * LPS>();
    else if (NS >= 3 && ahead == 1) wait_dma_and_barrier<LPS>();
    else wait_dma_and_barrier<0>();
    SM_T(unsigned long long st1 = sm_stamp(); tw += st1 - st0;)
    if (kt + NS - 1 < nkt) stage(kt + NS - 1, fill);
    SM_T(unsigned long long st2 = sm_stamp(); ti += st2 - st1;)
    const char* As = smem + cur * STAGE;
    const char* Ms = As + SA;
    const char* Bs = Ms + SM_;
    smfmac_stage<FM, FN, BF>(As, Ms, Bs, wm * TM, wn * TN, lane, acc);
    cur = cur + 1 == NS ? 0 : cur + 1;
    fill = fill + 1 == NS ? 0 : fill + 1;
    SM_T(__builtin_amdgcn_sched_barrier(0); st0 = sm_stamp(); tc += st0 - st2;)
  }
  __syncthreads();  // nothing is in flight here: the last NS-1 iterations issued no DMA
  SM_T(const unsigned long long tloop = sm_stamp();)

  store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, true, wm * TM, wn * TN, m0, n0, p.Mrows, p.N, p.alpha, p.beta, tid);
#ifdef SM_STAMP
  if (p.dbg && lane == 0) {
    const unsigned long long tend = sm_stamp();
    unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
    d[0] = tw; d[1] = ti; d[2] = tc; d[3] = tloop - tstart; d[4] = tend - tloop; d[5] = tstart; d[6] = tend;
    d[7] = __builtin_amdgcn_s_getreg(GETREG_IMMED(4 - 1, 0, 20 /*HW_REG_XCC_ID*/));
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Producer / consumer form of the same pipeline.  Cycle stamps of the kernel above (diagnostic build,
// profiles/stamp_r01.txt) show a wave parked for 500-1800 cycles per stage in the ISSUE of its
// global_load_lds instructions -- the vector-memory queue accepts them only as fast as L2 / HBM
// delivers -- and only then starting its SMFMACs: transfer and compute were serialised per wave.
// Here NL dedicated loader waves do nothing but issue the DMA (they are the ones that park), and the
// WM x WN consumer waves do nothing but LDS reads + SMFMAC.  One s_barrier per stage joins them:
//   loader  : wait vmcnt -> stage kt landed | barrier kt | issue stage kt+NS-1 (into the buffer the
//             consumers finished before barrier kt)
//   consumer:                                 barrier kt | compute stage kt
// so the transfer of stages kt+1 .. kt+NS-1 runs under the compute of stage kt.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NL, int NS, bool BF = false>
__global__ __launch_bounds__(64 * (WM * WN + NL)) void spmma_f16_pc_kernel(const SpmmaArgs p) {
  constexpr int NC = WM * WN, NW = NC + NL;
  static_assert(NS >= 2 && NS <= 6, "ring depth");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(FM >= 1 && FN >= 1, "wave tile");
  constexpr int SA = BM * 64, SM_ = BM * 8, SB = 64 * BN * 2, STAGE = SA + SM_ + SB;
  static_assert(BM % 128 == 0, "metadata DMA moves 128 rows per instruction");
  constexpr int A_N = BM / 16, M_N = BM / 128, B_N = BN / 8, W = A_N + M_N + B_N;
  constexpr int SL = (W + NL - 1) / NL;  // DMA slots per loader wave
  constexpr int LPS = W / NL;            // least any loader wave issues per stage: the vmcnt unit (a wave
                                         // with one more then waits slightly longer than it must -- safe)
  static_assert(LPS >= 1, "loader waves");
  static_assert((NS - 2) * ((W + NL - 1) / NL) <= 63, "the counted vmcnt must fit its 6-bit field");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned gb = lid / tiles, trem = lid - gb * tiles;
  const unsigned grp = gb / (unsigned)p.batch, b = gb - grp * (unsigned)p.batch;  // problem of the group, grid batch in it
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = p.kc / 64;
  half_t* C = p.C[grp] + (size_t)b * p.sC;

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned wm = wave / WN, wn = wave % WN;  // meaningful for consumer waves only

  if (wave >= (unsigned)NC) {
    // ------------------------------------------------------------------ loader wave
    const unsigned lw = wave - NC;
    const size_t row_base = (size_t)b * p.m;
    const char* vals = p.vals[grp] + row_base * 64;  // plane of stage 0; + Mtot * 64 per stage (stage-major values)
    const char* meta = p.meta[grp] + row_base * 8;
    const half_t* B = p.B[grp] + (size_t)b * p.sB;
    const int mlast = p.Mrows - 1;
    const char* src[SL];
    size_t step[SL];
    unsigned loff[SL];
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const unsigned t = lw + (unsigned)NL * i;
      if (t < (unsigned)A_N) {
        const unsigned row = 16u * t + (lane >> 2), cs = (lane & 3u) ^ a64_swz(row);
        int gr = m0 + (int)row;
        gr = gr < mlast ? gr : mlast;
        src[i] = vals + (size_t)gr * 64 + 16u * cs;
        step[i] = p.Mtot * 64;
        loff[i] = t * 1024u;
      } else if (t < (unsigned)(A_N + M_N)) {
        const unsigned u = t - A_N;
        size_t off = ((size_t)m0 + 128u * u + 2u * lane) * 8;
        const size_t last = (size_t)p.Mrows * 8 - 16;
        off = off < last ? off : last;
        src[i] = meta + off;
        step[i] = p.Mtot * 8;
        loff[i] = SA + u * 1024u;
      } else {
        const unsigned j = t - (A_N + M_N), panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
        const unsigned cs = (lane & 7u) ^ b_swz(kr);
        int gc = n0 + (int)(64u * panel + 8u * cs);
        gc = gc <= p.N - 8 ? gc : p.N - 8;
        src[i] = reinterpret_cast<const char*>(B + (size_t)kr * p.N + gc);
        step[i] = (size_t)64 * p.N * 2;
        loff[i] = SA + SM_ + panel * 8192u + (j & 7u) * 1024u;
      }
    }
    const bool ktail = (p.K & 63) != 0;
    auto stage = [&](int kt, int buf) {
      char* base = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < SL; ++i) {
        const unsigned t = lw + (unsigned)NL * i;  // wave-uniform
        if (t >= (unsigned)W) continue;
        gptr_t* gp = (gptr_t*)(src[i] + (size_t)kt * step[i]);
        lptr_t* lp = (lptr_t*)(base + loff[i]);
        if (ktail && kt == nkt - 1 && t >= (unsigned)(A_N + M_N)) {  // B rows of the K tail: zeros
          const unsigned kr = 8u * ((t - (unsigned)(A_N + M_N)) & 7u) + (lane >> 3);
          if (kt * 64 + (int)kr >= p.K) gp = (gptr_t*)(sm_zero_page + 16u * (lane & 7u));
        }
#ifdef SM_ABLATE  /* diagnostic timing builds only: 1 = no metadata DMA, 2 = no A DMA, 4 = no B DMA */
        if ((SM_ABLATE & 1) && t >= (unsigned)A_N && t < (unsigned)(A_N + M_N)) continue;
        if ((SM_ABLATE & 2) && t < (unsigned)A_N) continue;
        if ((SM_ABLATE & 4) && t >= (unsigned)(A_N + M_N)) continue;
#endif
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
      }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nkt) stage(s, s);
    int fill = NS - 1;
    SM_T(unsigned long long tw = 0, tb = 0, ti = 0; unsigned long long s0 = sm_stamp();)
    for (int kt = 0; kt < nkt; ++kt) {
      const int ahead = (nkt - 1 - kt) < (NS - 2) ? (nkt - 1 - kt) : (NS - 2);
#ifdef SM_STAMP
      if (NS >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
      else if (NS >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long s1 = sm_stamp(); tw += s1 - s0;
      asm volatile("s_barrier" ::: "memory");
      unsigned long long s2 = sm_stamp(); tb += s2 - s1;
#else
      if (NS >= 6 && ahead == 4) wait_dma_and_barrier<4 * LPS>();
      else if (NS >= 5 && ahead == 3) wait_dma_and_barrier<3 * LPS>();
      else if (NS >= 4 && ahead == 2) wait_dma_and_barrier<2 * LPS>();
      else if (NS >= 3 && ahead == 1) wait_dma_and_barrier<LPS>();
      else wait_dma_and_barrier<0>();
#endif
      if (kt + NS - 1 < nkt) stage(kt + NS - 1, fill);
      fill = fill + 1 == NS ? 0 : fill + 1;
      SM_T(s0 = sm_stamp(); ti += s0 - s2;)
    }
    SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8; d[0] = tw; d[1] = tb; d[2] = ti; d[3] = 1; })
  } else {
    // ------------------------------------------------------------------ consumer wave
    int cur = 0;
    SM_T(unsigned long long tb = 0, tc = 0; unsigned long long s0 = sm_stamp();)
    for (int kt = 0; kt < nkt; ++kt) {
      wait_dma_and_barrier<0>();  // a consumer has no DMA of its own: this is the stage barrier
      SM_T(unsigned long long s1 = sm_stamp(); tb += s1 - s0;)
      const char* As = smem + cur * STAGE;
      const char* Ms = As + SA;
      const char* Bs = Ms + SM_;
      smfmac_stage<FM, FN, BF>(As, Ms, Bs, wm * TM, wn * TN, lane, acc);
      cur = cur + 1 == NS ? 0 : cur + 1;
      SM_T(__builtin_amdgcn_sched_barrier(0); s0 = sm_stamp(); tc += s0 - s1;)
    }
    SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8; d[0] = 0; d[1] = tb; d[2] = tc; d[3] = 0; })
  }
  __syncthreads();  // both roles; nothing is in flight (the last NS-1 loader iterations issued no DMA)

  // ---- epilogue: consumers stage their fragments, every wave (loaders too) stores 16-byte row pieces
  store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, wave < (unsigned)NC, wm * TM, wn * TN, m0, n0, p.Mrows, p.N, p.alpha, p.beta, tid);
}

// (A variant of this kernel with the consumers' operand reads pipelined across stages -- ring of 4, the next stage's A
// fragments and first B fragment fetched under the current stage's last SMFMACs, B fragments two ahead -- was built and
// measured in round 2: bit-identical, and 1.4 x SLOWER on every few-tile shape (196x512x4608: 51 vs 37 us,
// profiles/tune_pc2_r02i.txt), with or without the deeper look-ahead.  Exposed LDS latency in the consumers is therefore
// not what bounds this kernel; the variant is in the git history, not in the library.)
template <int BM, int BN, int WM, int WN, int NL, int NS, bool BF = false>
static int launch_pc(const SpmmaArgs& a0, hipStream_t st) {
  SpmmaArgs a = a0;
  a.tiles_m = (a.Mrows + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch * a.ngroup;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("spmma_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = NS * ((size_t)BM * 72 + (size_t)64 * BN * 2);
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;  // the most this configuration ever asks for
  // a K of fewer stages than the ring never touches the ring's other buffers: without them more workgroups share a CU
  const size_t used = ((size_t)(a.kc / 64) < (size_t)NS ? (size_t)(a.kc / 64) : (size_t)NS) * ((size_t)BM * 72 + (size_t)64 * BN * 2);
  const size_t lds_launch = used > lds_epi ? used : lds_epi;
  static LdsOptIn lds_optin;
  if (lds > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_pc_kernel<BM, BN, WM, WN, NL, NS, BF>), lds, "spmma_f16_pc_kernel")) return rc;
  }
#ifdef SM_STAMP
  {
    static unsigned long long* dbg = nullptr;
    constexpr int NWV = WM * WN + NL;
    const size_t cnt = nwg * (size_t)NWV * 8;
    static size_t cap = 0;
    if (cnt > cap) { if (dbg) (void)hipFree(dbg); (void)hipMalloc((void**)&dbg, cnt * 8); cap = cnt; }
    a.dbg = dbg;
    spmma_f16_pc_kernel<BM, BN, WM, WN, NL, NS, BF><<<dim3((unsigned)nwg), dim3(64 * NWV), lds_launch, st>>>(a);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(cnt);
    (void)hipMemcpy(h.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
    double L[3] = {0, 0, 0}, Cn[2] = {0, 0};
    double nl = 0, nc = 0;
    for (size_t i = 0; i < cnt / 8; ++i) {
      if (h[i * 8 + 3] == 1) { L[0] += h[i * 8]; L[1] += h[i * 8 + 1]; L[2] += h[i * 8 + 2]; nl += 1; }
      else { Cn[0] += h[i * 8 + 1]; Cn[1] += h[i * 8 + 2]; nc += 1; }
    }
    const double nk = (double)(a.kc / 64);
    fprintf(stderr, "STAMP-PC %dx%dx%d NL=%d NS=%d tiles=%zu nkt=%d | loader per stage: vmcnt-wait %.0f barrier %.0f issue %.0f | consumer per stage: barrier %.0f compute %.0f\n",
            a.Mrows, a.N, a.K, NL, NS, nwg, a.kc / 64, L[0] / nl / nk, L[1] / nl / nk, L[2] / nl / nk, Cn[0] / nc / nk, Cn[1] / nc / nk);
    return check_launch("spmma_f16_pc_kernel");
  }
#endif
  spmma_f16_pc_kernel<BM, BN, WM, WN, NL, NS, BF><<<dim3((unsigned)nwg), dim3(64 * (WM * WN + NL)), lds_launch, st>>>(a);
  return check_launch("spmma_f16_pc_kernel");
}

template <int BM, int BN, int WM, int WN, int NS, bool BF = false>
static int launch_dma(const SpmmaArgs& a0, hipStream_t st) {
  SpmmaArgs a = a0;
  a.tiles_m = (a.Mrows + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch * a.ngroup;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("spmma_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = NS * ((size_t)BM * 72 + (size_t)64 * BN * 2);
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;  // the most this configuration ever asks for
  // a K of fewer stages than the ring never touches the ring's other buffers: without them more workgroups share a CU
  const size_t used = ((size_t)(a.kc / 64) < (size_t)NS ? (size_t)(a.kc / 64) : (size_t)NS) * ((size_t)BM * 72 + (size_t)64 * BN * 2);
  const size_t lds_launch = used > lds_epi ? used : lds_epi;
  static LdsOptIn lds_optin;
  if (lds > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_dma_kernel<BM, BN, WM, WN, NS, BF>), lds, "spmma_f16_dma_kernel")) return rc;
  }
#ifdef SM_STAMP
  {
    static unsigned long long* dbg = nullptr;
    const size_t cnt = nwg * (size_t)(WM * WN) * 8;
    static size_t cap = 0;
    if (cnt > cap) { if (dbg) (void)hipFree(dbg); (void)hipMalloc((void**)&dbg, cnt * 8); cap = cnt; }
    a.dbg = dbg;
    spmma_f16_dma_kernel<BM, BN, WM, WN, NS, BF><<<dim3((unsigned)nwg), dim3(64 * WM * WN), lds_launch, st>>>(a);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(cnt);
    (void)hipMemcpy(h.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
    double s[5] = {0, 0, 0, 0, 0};
    unsigned long long t_min = ~0ull, t_max = 0;
    for (size_t i = 0; i < cnt / 8; ++i) {
      for (int j = 0; j < 5; ++j) s[j] += (double)h[i * 8 + j];
      if (h[i * 8 + 5] < t_min) t_min = h[i * 8 + 5];
      if (h[i * 8 + 6] > t_max) t_max = h[i * 8 + 6];
    }
    const double nwv = (double)(cnt / 8), nk = (double)(a.kc / 64);
    fprintf(stderr, "STAMP %dx%dx%d nw=%d ns=%d tiles=%zu nkt=%d | per wave per stage: wait %.0f issue %.0f compute %.0f | loop %.0f epilogue %.0f cycles | kernel span %.0f cycles (100MHz ticks? no: shader clk)\n",
            a.Mrows, a.N, a.K, WM * WN, NS, nwg, a.kc / 64, s[0] / nwv / nk, s[1] / nwv / nk, s[2] / nwv / nk, s[3] / nwv, s[4] / nwv,
            (double)(t_max - t_min));
    return check_launch("spmma_f16_dma_kernel");
  }
#endif
  spmma_f16_dma_kernel<BM, BN, WM, WN, NS, BF><<<dim3((unsigned)nwg), dim3(64 * WM * WN), lds_launch, st>>>(a);
  return check_launch("spmma_f16_dma_kernel");
}

template <int BM, int BN, int WM, int WN, bool BF = false>
static int launch_cfg(const SpmmaArgs& a0, hipStream_t st) {
  SpmmaArgs a = a0;
  a.tiles_m = (a.Mrows + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch * a.ngroup;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("spmma_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = (size_t)BM * 144 + (size_t)(BN / 64) * 128 * 128;
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  spmma_f16_kernel<BM, BN, WM, WN, BF><<<dim3((unsigned)nwg), dim3(256), lds, st>>>(a);
  return check_launch("spmma_f16_kernel");
}

}  // namespace sm

using namespace sm;

// BF = false: fp16, true: bfloat16 -- same blob layout, same kernels, other matrix instruction and final rounding.
// ng <= SPMMA_MAXG same-shape problems (blob, B, C triples) as one grid; a plain call is a group of one.
template <bool BF>
static int spmma16(size_t ng, const void* const* blobs, const void* const* Bs, void* const* Cs, size_t m, size_t n, size_t k, size_t batch,
                   size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream) {
  if (ng == 0) return SM_STATUS_SUCCESS;
  if (!blobs || !Bs || !Cs || ng > (size_t)SPMMA_MAXG) {
    set_error("sm_spmma_{f16,bf16}: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  bool b_aligned = true;
  for (size_t g = 0; g < ng; ++g) {
    if (!blobs[g] || !Bs[g] || !Cs[g] || !aligned16(blobs[g])) {
      set_error("sm_spmma_{f16,bf16}: invalid argument (blob must be 16-byte aligned)");
      return SM_STATUS_INVALID_VALUE;
    }
    b_aligned = b_aligned && aligned16(Bs[g]);
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull) {
    set_error("sm_spmma_{f16,bf16}: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  const BlobLayout L = blob_layout(m, k, 2, batch);
  SpmmaArgs a = {};
  for (size_t g = 0; g < (size_t)SPMMA_MAXG; ++g) {  // unused slots repeat problem 0 (never indexed: grp < ngroup)
    const size_t s_ = g < ng ? g : 0;
    a.vals[g] = (const char*)blobs[s_];
    a.meta[g] = (const char*)blobs[s_] + L.meta_off;
    a.B[g] = (const half_t*)Bs[s_];
    a.C[g] = (half_t*)Cs[s_];
  }
  a.ngroup = (int)ng;
  a.Mtot = L.M;
  a.sB = strideB; a.sC = strideC;
  a.m = (int)m; a.Mrows = (int)m; a.N = (int)n; a.K = (int)k; a.kc = (int)L.kc;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  // shared B + contiguous C: the batch is one tall matrix of batch*m blob rows
  if (batch > 1 && strideB == 0 && strideC == m * n) {
    a.Mrows = (int)(m * batch);
    a.batch = 1;
  }
  hipStream_t st = (hipStream_t)stream;
  // the metadata DMA moves 16-byte row pairs: batches and planes must start on even rows
  const bool fast = (n % 8 == 0) && n >= 8 && b_aligned && (strideB % 8 == 0) && (m % 2 == 0) && k >= 8;
  if (fast) {
    // Workgroup shape by how many tiles exist: with thousands of tiles 4 waves per tile and several
    // tiles per CU overlap each other's latencies; with about one tile per CU the same tile is spread
    // over 8 or 16 waves so that every SIMD still holds several waves.  SM_SPMMA_CFG (tuning aid):
    // "<waves>x<ring>" forces a configuration.
    // 256 x 128 tiles (B lines amortised over twice the rows) pay with a long K and enough rows for >= 64 such
    // tiles per n-tile (profiles/sweep_r01_*.txt: 784x256x{1024,2304}, 3136x128x1152 at b=32)
    if (!tuning_env("SM_SPMMA_PC") && !tuning_env("SM_SPMMA_CFG") && n >= 128 && n <= 256 && k >= 1024 &&
        (size_t)a.Mrows * ng >= 16384)
      return launch_pc<256, 128, 4, 2, 4, 3, BF>(a, st);
#ifdef SM_TUNING
    {  // A/B: 256 x 256 tiles, eight waves of 32 rows x 256 columns, all waves issue the DMA (ring of 3 / 2): SM_SPMMA_BIG = 3 / 2
      const int big = tuning_int("SM_SPMMA_BIG", 0);
      if (big && n > 128 && k > 64) return big == 2 ? launch_dma<256, 256, 8, 1, 2, BF>(a, st) : launch_dma<256, 256, 8, 1, 3, BF>(a, st);
    }
#endif
    static const char* pc_env = tuning_env("SM_SPMMA_PC");  // tuning aid: "<loaders>x<ring>", "0" = previous kernel
    // default: long K -> producer/consumer kernel (4 loader waves, ring of 3); short K -> the kernel
    // above with more tiles per CU (measured per shape on the ResNet tables, profiles/sweep_r01_*.txt)
    // (n <= 64 is HBM-bound at every K: the plain DMA kernel with more tiles per CU wins there)
    int nl = (k >= 512 && n > 64) ? 4 : 0, pns = 3;
    if (pc_env) sscanf(pc_env, "%dx%d", &nl, &pns);
    if (nl == 256 && n > 64) {  // tuning aid: 256 x 128 tiles, 8 consumer waves (64 x 64) + 4 loaders, 64-deep stages
      return pns >= 4 ? launch_pc<256, 128, 4, 2, 4, 4, BF>(a, st) : (pns == 3 ? launch_pc<256, 128, 4, 2, 4, 3, BF>(a, st) : launch_pc<256, 128, 4, 2, 4, 2, BF>(a, st));
    }
#ifdef SM_TUNING
    if (pc_env && n > 64) {  // deep rings / more loaders / 8 consumer waves on 128 x 128 tiles: "<nl>x<ns>x<c>", c = 4 or 8 consumers
      int cw = 4;
      sscanf(pc_env, "%*dx%*dx%d", &cw);
      if (cw == 8) {
        if (nl == 8) return pns >= 5 ? launch_pc<128, 128, 2, 4, 8, 5, BF>(a, st) : (pns == 4 ? launch_pc<128, 128, 2, 4, 8, 4, BF>(a, st) : launch_pc<128, 128, 2, 4, 8, 3, BF>(a, st));
        return pns >= 5 ? launch_pc<128, 128, 2, 4, 4, 5, BF>(a, st) : (pns == 4 ? launch_pc<128, 128, 2, 4, 4, 4, BF>(a, st) : launch_pc<128, 128, 2, 4, 4, 3, BF>(a, st));
      }
      if (nl == 8) return pns >= 6 ? launch_pc<128, 128, 2, 2, 8, 6, BF>(a, st) : (pns == 5 ? launch_pc<128, 128, 2, 2, 8, 5, BF>(a, st) : (pns == 4 ? launch_pc<128, 128, 2, 2, 8, 4, BF>(a, st) : launch_pc<128, 128, 2, 2, 8, 3, BF>(a, st)));
      if (nl == 4 && pns >= 5) return pns >= 6 ? launch_pc<128, 128, 2, 2, 4, 6, BF>(a, st) : launch_pc<128, 128, 2, 2, 4, 5, BF>(a, st);
    }
#endif
    if (nl > 0) {
      if (n <= 64) {
        if (nl == 2) return pns >= 3 ? launch_pc<128, 64, 4, 1, 2, 3, BF>(a, st) : launch_pc<128, 64, 4, 1, 2, 2, BF>(a, st);
        return pns >= 4 ? launch_pc<128, 64, 4, 1, 4, 4, BF>(a, st) : (pns == 3 ? launch_pc<128, 64, 4, 1, 4, 3, BF>(a, st) : launch_pc<128, 64, 4, 1, 4, 2, BF>(a, st));
      }
      if (nl == 2) return pns >= 3 ? launch_pc<128, 128, 2, 2, 2, 3, BF>(a, st) : launch_pc<128, 128, 2, 2, 2, 2, BF>(a, st);
      return pns >= 4 ? launch_pc<128, 128, 2, 2, 4, 4, BF>(a, st) : (pns == 3 ? launch_pc<128, 128, 2, 2, 4, 3, BF>(a, st) : launch_pc<128, 128, 2, 2, 4, 2, BF>(a, st));
    }
    static const char* cfg_env = tuning_env("SM_SPMMA_CFG");
    const size_t Mr = (size_t)a.Mrows;
    int nw, ns;
    if (n <= 64) {
      const size_t tiles = ceil_div(Mr, 128) * a.batch * ng;
      nw = tiles >= 1024 ? 4 : 8;
      ns = 2;
      if (cfg_env) sscanf(cfg_env, "%dx%d", &nw, &ns);
      if (nw >= 8) return ns >= 3 ? launch_dma<128, 64, 4, 2, 3, BF>(a, st) : launch_dma<128, 64, 4, 2, 2, BF>(a, st);
      return ns >= 3 ? launch_dma<128, 64, 4, 1, 3, BF>(a, st) : launch_dma<128, 64, 4, 1, 2, BF>(a, st);
    }
    const size_t tiles = ceil_div(Mr, 128) * ceil_div(n, 128) * a.batch * ng;
    nw = tiles >= 1024 ? 4 : (tiles >= 512 ? 8 : 16);
    ns = (tiles < 512 && k >= 1024) ? 3 : 2;
    if (cfg_env) sscanf(cfg_env, "%dx%d", &nw, &ns);
    if (nw >= 16) return ns >= 3 ? launch_dma<128, 128, 4, 4, 3, BF>(a, st) : launch_dma<128, 128, 4, 4, 2, BF>(a, st);
    if (nw >= 8) return ns >= 3 ? launch_dma<128, 128, 2, 4, 3, BF>(a, st) : launch_dma<128, 128, 2, 4, 2, BF>(a, st);
    return ns >= 3 ? launch_dma<128, 128, 2, 2, 3, BF>(a, st) : launch_dma<128, 128, 2, 2, 2, BF>(a, st);
  }
  if (n <= 64) return launch_cfg<128, 64, 4, 1, BF>(a, st);
  return launch_cfg<128, 128, 2, 2, BF>(a, st);
}

extern "C" int sm_spmma_f16(const void* blob, const void* B, void* C, size_t m, size_t n, size_t k, size_t batch,
                            size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream) {
  return spmma16<false>(1, &blob, &B, &C, m, n, k, batch, strideB, strideC, alpha, beta, stream);
}
extern "C" int sm_spmma_bf16(const void* blob, const void* B, void* C, size_t m, size_t n, size_t k, size_t batch,
                             size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream) {
  return spmma16<true>(1, &blob, &B, &C, m, n, k, batch, strideB, strideC, alpha, beta, stream);
}

// `count` same-shape problems in as few grids as possible (SPMMA_MAXG per launch): same kernels, same C bit for bit
template <bool BF>
static int spmma16_grouped(size_t count, const void* const* blobs, const void* const* Bs, void* const* Cs, size_t m, size_t n, size_t k, size_t batch,
                           size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream) {
  if (count == 0) return SM_STATUS_SUCCESS;
  if (!blobs || !Bs || !Cs) {
    set_error("sm_spmma_{f16,bf16}_grouped: null pointer table");
    return SM_STATUS_INVALID_VALUE;
  }
  for (size_t i = 0; i < count; i += (size_t)SPMMA_MAXG) {
    const size_t ng = count - i < (size_t)SPMMA_MAXG ? count - i : (size_t)SPMMA_MAXG;
    if (const int rc = spmma16<BF>(ng, blobs + i, Bs + i, Cs + i, m, n, k, batch, strideB, strideC, alpha, beta, stream)) return rc;
  }
  return SM_STATUS_SUCCESS;
}
extern "C" int sm_spmma_f16_grouped(size_t count, const void* const* blobs, const void* const* B, void* const* C, size_t m, size_t n, size_t k,
                                    size_t batch, size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream) {
  return spmma16_grouped<false>(count, blobs, B, C, m, n, k, batch, strideB, strideC, alpha, beta, stream);
}
extern "C" int sm_spmma_bf16_grouped(size_t count, const void* const* blobs, const void* const* B, void* const* C, size_t m, size_t n, size_t k,
                                     size_t batch, size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream) {
  return spmma16_grouped<true>(count, blobs, B, C, m, n, k, batch, strideB, strideC, alpha, beta, stream);
}
