// spmma_f16_fused.hip -- fused prune -> compress -> 2:4 matmul (row f-1 of the hot-path table: "fused
// prune->compress->SpMMA pipeline"; the reference re-creates and re-reads a compressed blob on every call,
// include/sparsify.me/spmma.hxx:86-113).  C = alpha * prune24_strip(A) * B + beta * C straight from the
// DENSE A: the 2:4 selection happens in the loader waves' registers and only the kept pair + its nibble
// ever reach LDS, so per call the dense A is read once (m*k*s bytes) and no blob is written or re-read
// (the staged path moves m*k*s + 2 * (m*k*s/2 + m*k/8) bytes for the same result).
//
// Same producer/consumer skeleton as spmma_f16.hip's pc kernel (64-deep stages, one barrier per stage):
//   A loader waves (4): plain 16-byte loads (8 rows x 128 B per wave instruction: whole cache lines), PF
//     stages ahead in registers; per 8 loaded halves (two strips) the STRIP rule (select24.h:
//     strip_select_f16) -> 4 kept halves + 1 metadata byte, written with ds_write_b64 / b8 into exactly
//     the LDS images the consumers of the staged kernel read (ring of 2);
//   B loader waves (1-4): the B tile by LDS-DMA into its own ring of NSB stages.  They are separate waves
//     because vmcnt retires in order per wave: a wave that waits for its B DMA also drains every older A
//     load, which would cap the A data in flight at one stage (16 KiB per CU: latency-bound);
//   consumer waves: unchanged (ds_read_b128 + ds_read_u16 + ds_read_b64_tr_b16 + v_smfmac).
// The result is bit-identical to sm_spmma_f16(sm_compress24_f16(A), B): same kept values, same codes, same
// instruction sequence on the same operands (tests/test_gpu_parity.py::test_fused_equals_staged).
#include <cstdio>
#include <vector>

#include "select24.h"
#include "spmma_args.h"

namespace sm {

// A launch serves up to MAXG same-shape problems (the grouped entry points: the 3-6 instances of one layer shape in a
// network run as ONE grid, so the tail of one instance is filled by the next and few-tile shapes stop paying whole
// rounds of 256 CUs per instance); problem g = (A[g], B[g], C[g]), every other field shared.  A plain call is ngroup = 1.
constexpr int MAXG = 8;
struct FusedArgs {
  const half_t* A[MAXG];
  const half_t* B[MAXG];
  half_t* C[MAXG];
  size_t sA, sB, sC;  // batch strides (elements)
  int Mrows, N, K, lda;
  int batch, tiles_m, tiles_n;
  int ngroup;
  float alpha, beta;
  // dense twin through sm_gemm_batched_*: DEVICE arrays of per-batch pointers (null: the tables above + batch strides)
  const half_t* const* dAp;
  const half_t* const* dBp;
  half_t* const* dCp;
#ifdef SM_STAMP
  unsigned long long* dbg;  // diagnostic build only: per-wave cycle sums (never in the product library)
#endif
#ifdef SM_TUNING
  int ablate;  // tuning builds only (direct kernel): 1 = no C store, 2 = no stage compute, 4 = no A / B loads (timing only: C is wrong)
#endif
};

// ---------------------------------------------------------------------------------------------
// n <= 128, DIRECT form: no loader waves and no compressed image at all.  The dense A tile (128 rows x 128 B per 64-k
// stage) and the B tile reach LDS by LDS-DMA issued by the four compute waves themselves (ring of NS stages, counted
// vmcnt, one barrier per stage: the structure of spmma_f16_dma_kernel); wave w owns rows 32w .. 32w+31 and ALL columns,
// so every row of A is selected exactly once -- by the lane that feeds it to the SMFMAC (smfmac_stage_dense_a).
// The data in flight are LDS buffers, not registers: 2 x 24 KiB per workgroup, two workgroups per CU at n = 64.
// ---------------------------------------------------------------------------------------------
// (round 5) the 128-column form is asked to fit four waves per SIMD (<= 128 registers; left alone it takes 129 = three): a single-stage
// problem (k = 64) allocates 34 KiB of LDS, so a fourth workgroup then shares the CU -- with no K loop to pipeline, the workgroups
// per CU are all that overlaps one tile's load latency with another's stores (12544 x 256 x 64)
// (measured indifferent: profiles/direct_ablate_occ4_r05p.txt -- kept, it costs nothing.  Tuning builds, whose ablation switches push the kernel past 128 registers,
// keep the default bound: with the hint they spill 204 bytes per lane and run 3 x slower, profiles/ab_bm192_r05s.txt)
#ifdef SM_TUNING
#define SM_DIRECT_MIN_WAVES(BN, NWV, BM) 1
#else
#define SM_DIRECT_MIN_WAVES(BN, NWV, BM) (((BN) == 128 && (NWV) == 4 && (BM) == 128) ? 4 : 1)
#endif
template <int BN, int NS, bool BF = false, int BM = 128, int NWV = 4, bool ANT = true, bool DENSE = false>
__global__ __launch_bounds__(64 * NWV, SM_DIRECT_MIN_WAVES(BN, NWV, BM)) void spmma_f16_fused_direct_kernel(const FusedArgs p) {
  static_assert(BM == 128 || BM == 64, "row tile");
  static_assert(NWV == 4 || (NWV == 8 && BM == 128), "waves per workgroup");
  constexpr int NW = NWV, TM = BM / NW, FM = TM / 16, FN = BN / 16;
  constexpr int SA = BM * 128, SB = 64 * BN * 2, STAGE = SA + SB;
  constexpr int A_N = BM / 8, B_N = BN / 8, W = A_N + B_N;  // 1 KiB DMA wave-instructions per stage
  static_assert(W % NW == 0, "equal DMA share per wave");
  constexpr int SL = W / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
#ifdef SM_TUNING
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, !(p.ablate & 16));  // (tuning: SM_DIRECT_ABLATE bit 4 = the XCD ranges of rounds 1-4)
#else
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, true);  // dispatch order: mma_tile.h
#endif
  const unsigned gb = lid / tiles, trem = lid - gb * tiles;
  const unsigned grp = gb / (unsigned)p.batch, b = gb - grp * (unsigned)p.batch;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = p.K / 64;
  const half_t* A = p.dAp ? p.dAp[gb] : p.A[grp] + (size_t)b * p.sA;  // (device pointer tables: the dense twin behind sm_gemm_batched_*)
  const half_t* B = p.dBp ? p.dBp[gb] : p.B[grp] + (size_t)b * p.sB;
  half_t* C = p.dCp ? p.dCp[gb] : p.C[grp] + (size_t)b * p.sC;
  const int mlast = p.Mrows - 1;
  // (Round 5: the ablation of this kernel -- tools/direct_ablate.py, profiles/direct_ablate_r05n.txt -- reads "whole launch = launch without
  //  the C store + the C store alone" on every shape (12544 x 256 x 64 x 3: 220 = 105 + 113 us): loads and stores do not overlap.  Starting the
  //  workgroups that share a CU a third of a tile's lifetime apart, so that one stores while the others load, changed nothing
  //  (profiles/direct_stagger_r05o.txt): it is not a phase alignment of the resident workgroups.  Removed again.)

  const char* src[SL];
  size_t step[SL];
  unsigned loff[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    if (t < (unsigned)A_N) {  // 8 rows x 128 B: lane -> row 8t + lane/8, LDS chunk lane%8 holds source chunk (lane%8) ^ (row&7)
      const unsigned row = 8u * t + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      src[i] = reinterpret_cast<const char*>(A + (size_t)gr * p.lda) + 16u * cs;
      step[i] = 128;
      loff[i] = t * 1024u;
    } else {
      const unsigned j = t - A_N, panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ b_swz(kr);
      int gc = n0 + (int)(64u * panel + 8u * cs);
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      src[i] = reinterpret_cast<const char*>(B + (size_t)kr * p.N + gc);
      step[i] = (size_t)64 * p.N * 2;
      loff[i] = SA + panel * 8192u + (j & 7u) * 1024u;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
#ifdef SM_TUNING
    if (p.ablate & 4) return;
#endif
    // A is read exactly once by the whole grid: its DMA carries the non-temporal hint (aux = 2, `nt`); B is re-read by
    // every row tile and keeps the default policy.  Instruction i of a wave is an A piece iff NW * i < A_N (A_N % NW == 0).
    static_assert(A_N % NW == 0, "A / B split of the DMA instructions is per instruction index");
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      if (ANT && NW * i < A_N)
        __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)kt * step[i]), (lptr_t*)(base + loff[i]), 16, 0, 2);
      else
        __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)kt * step[i]), (lptr_t*)(base + loff[i]), 16, 0, 0);
    }
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) stage(s, s);
  int cur = 0, fill = NS - 1;
  SM_T(unsigned long long tv = 0, tb = 0, ti = 0, tc = 0; unsigned long long s0 = sm_stamp(); const unsigned long long sstart = s0;)
  for (int kt = 0; kt < nkt; ++kt) {
    const int ahead = (nkt - 1 - kt) < (NS - 2) ? (nkt - 1 - kt) : (NS - 2);
#ifdef SM_STAMP
    if (NS >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SL) : "memory");
    else if (NS >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long sv = sm_stamp(); tv += sv - s0;
    asm volatile("s_barrier" ::: "memory");
    const unsigned long long sb = sm_stamp(); tb += sb - sv;
#else
    if (NS >= 4 && ahead == 2) wait_dma_and_barrier<2 * SL>();
    else if (NS >= 3 && ahead == 1) wait_dma_and_barrier<SL>();
    else wait_dma_and_barrier<0>();
#endif
    if (kt + NS - 1 < nkt) stage(kt + NS - 1, fill);
    SM_T(const unsigned long long si = sm_stamp(); ti += si - sb;)
    const char* As = smem + cur * STAGE;
#ifdef SM_TUNING
    if (!(p.ablate & 2)) {
#endif
    if constexpr (DENSE) mfma_stage_dense_a<FM, FN, BF>(As, As + SA, wave * TM, 0, lane, acc);  // the dense twin: every element multiplied
    else smfmac_stage_dense_a<FM, FN, BF>(As, As + SA, wave * TM, 0, lane, acc);
#ifdef SM_TUNING
    }
#endif
    cur = cur + 1 == NS ? 0 : cur + 1;
    fill = fill + 1 == NS ? 0 : fill + 1;
    SM_T(__builtin_amdgcn_sched_barrier(0); s0 = sm_stamp(); tc += s0 - si;)
  }
  __syncthreads();
  SM_T(const unsigned long long sloop = sm_stamp();)
#ifdef SM_TUNING
  if (p.ablate & 1) {  // no C store: one element per workgroup keeps the accumulators alive
    if (tid == 0) C[(size_t)m0 * p.N + n0] = to_elt<BF>(acc[0][0][0] + acc[FM - 1][FN - 1][3]);
    return;
  }
#endif
  store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, true, wave * TM, 0, m0, n0, p.Mrows, p.N, p.alpha, p.beta, tid);
  SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8; const unsigned long long se = sm_stamp();
        d[0] = tv; d[1] = tb; d[2] = ti; d[3] = tc; d[4] = sloop - sstart; d[5] = se - sloop; })
}

template <int BN, int NS, bool BF = false, int BM = 128, int NWV = 4, bool ANT = true, bool DENSE = false>
static int launch_fused_direct(const FusedArgs& a0, hipStream_t st) {
  FusedArgs a = a0;
#ifdef SM_TUNING
  a.ablate = tuning_int("SM_DIRECT_ABLATE", 0);
#endif
  a.tiles_m = (a.Mrows + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch * a.ngroup;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_spmma_fused_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  // a single-stage problem (k = 64) never touches the ring's other buffers: without them more workgroups share a CU
  constexpr size_t stage_bytes = BM * 128 + 64 * BN * 2;
  const size_t lds_main = ((size_t)(a.K / 64) < (size_t)NS ? (size_t)(a.K / 64) : (size_t)NS) * stage_bytes;
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  // the opt-in records the most this instantiation can ever ask for, the launch below only what this K touches
  constexpr size_t lds_max = NS * stage_bytes > lds_epi ? NS * stage_bytes : lds_epi;
  static LdsOptIn lds_optin;
  if (lds_max > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_fused_direct_kernel<BN, NS, BF, BM, NWV, ANT, DENSE>), lds_max, "spmma_f16_fused_direct_kernel")) return rc;
  }
#ifdef SM_STAMP
  {
    static unsigned long long* dbg = nullptr;
    static size_t cap = 0;
    const size_t cnt = nwg * NWV * 8;
    if (cnt > cap) { if (dbg) (void)hipFree(dbg); (void)hipMalloc((void**)&dbg, cnt * 8); cap = cnt; }
    (void)hipMemset(dbg, 0, cnt * 8);
    a.dbg = dbg;
    spmma_f16_fused_direct_kernel<BN, NS, BF, BM, NWV, ANT, DENSE><<<dim3((unsigned)nwg), dim3(64 * NWV), lds, st>>>(a);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(cnt);
    (void)hipMemcpy(h.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
    double t[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < cnt / 8; ++i)
      for (int j = 0; j < 6; ++j) t[j] += (double)h[i * 8 + j];
    const double nwv = (double)(cnt / 8), nk = (double)(a.K / 64);
    fprintf(stderr, "STAMP-FUSED-DIRECT %dx%dx%d NS=%d tiles=%zu nkt=%d | per wave per stage: vmcnt-wait %.0f barrier %.0f dma-issue %.0f compute %.0f | loop %.0f epilogue %.0f cycles per tile\n",
            a.Mrows, a.N, a.K, NS, nwg, a.K / 64, t[0] / nwv / nk, t[1] / nwv / nk, t[2] / nwv / nk, t[3] / nwv / nk, t[4] / nwv, t[5] / nwv);
    return check_launch("spmma_f16_fused_direct_kernel");
  }
#endif
  spmma_f16_fused_direct_kernel<BN, NS, BF, BM, NWV, ANT, DENSE><<<dim3((unsigned)nwg), dim3(64 * NWV), lds, st>>>(a);
  return check_launch("spmma_f16_fused_direct_kernel");
}

// ---------------------------------------------------------------------------------------------
// BIG form of the direct kernel (round 4): 256-row tiles, eight waves, split rings.  Same ownership as the direct kernel
// -- wave w owns rows 32 w .. 32 w + 31 of the tile and ALL its BN columns, every row of A is selected exactly once, by
// the lane that feeds it to the SMFMAC -- but one workgroup per CU with the whole LDS: the dense A stages (256 rows x
// 128 B = 32 KiB) sit in a ring of NSA = 3 and the B stages (64 x BN halves) in a ring of NSB = 2, so TWO stages of A
// (the operand that comes from HBM) are in flight per CU while B (served by L2) needs one.  Why: the few-tile families
//  are bound by the per-stage synchronisation chain at 21-47 lines outstanding per CU; a 256 x 256 tile
// does twice the work per barrier, halves the B bytes a CU pulls from L2 per A byte (A : B = 1 : 1 instead of 1 : 2) and
// keeps 64 KiB of HBM reads in flight per CU.  Issue order per iteration: B(kt + NSB - 1) then A(kt + NSA - 1); vmcnt
// retires in order, so at iteration kt's wait everything up to B(kt) has landed once only the pieces issued after it
// -- A(kt + 1) when NSA = 3 / NSB = 2 -- remain.  Same operands, same instruction sequence per output element as the
// direct / wide kernels: bit-identical C.
// Where a stage's ~3 900 cycles go (s_memtime stamps, profiles/stamp_r04b_big_wide_direct.txt; per wave): vmcnt wait 280, barrier
// 745, DMA issue 800 (eight 1 KiB pieces), selection 990, B sweep 1 100 -- the sweep is the LDS array's rate (eight waves each
// read the whole 32 KiB B stage = 1 024 LDS cycles), the A share of a stage (32 KiB) is 73 % of a CU's fair share of the HBM
// rate.  Built, bit-identical and NOT adopted (git history, profiles/ab_pingpong_r04h.txt, ab_variants_r04c.txt): two B
// fragments in flight instead of one (no change: the sweep is throughput-, not latency-bound), waves 4-7 issuing their A
// pieces after their compute (-1 .. +5 %), a ping-pong form in which waves 0-3 select one stage ahead so that one wave
// of every SIMD is on the VALU while its partner is on the matrix pipe (+0 .. +4 % time), and the stage's eight DMA pieces
// per wave spread over the B sweep -- one after every second fragment's SMFMACs -- instead of issued in one burst behind the
// barrier (profiles/ab_ilv_r04n.txt: -3 .. +3 % over seven shapes, A pieces only: -3 .. 0 %: the 800 cycles the burst spends in
// issue are not won back by hiding them, so they are not what bounds the stage).
// ---------------------------------------------------------------------------------------------
template <int BN, bool BF = false, int NSA = 3, int NSB = 2, bool ANT = true, bool DENSE = false>
__global__ __launch_bounds__(512) void spmma_f16_fused_big_kernel(const FusedArgs p) {
  constexpr int BM = 256, NW = 8, TM = BM / NW, FM = TM / 16, FN = BN / 16;
  constexpr int SA = BM * 128, SB = 64 * BN * 2;
  constexpr int A_N = BM / 8, B_N = BN / 8;  // 1 KiB DMA wave-instructions per stage
  static_assert(A_N % NW == 0 && B_N % NW == 0, "equal DMA share per wave");
  static_assert(NSA == 3 && (NSB == 2 || NSB == 3), "ring depths the counted waits below are written for");
  constexpr int SLA = A_N / NW, SLB = B_N / NW;
  constexpr int AHEAD = SLA + (NSB == 3 ? SLB : 0);  // pieces issued after B(kt) / A(kt) that may stay in flight at iteration kt
  constexpr int BRING = NSA * SA;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
#ifdef SM_TUNING
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, (p.ablate & 8) != 0);  // (tuning: SM_DIRECT_ABLATE bit 3 = dispatch order; measured 0-5 % slower here, mma_tile.h)
#else
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
#endif
  const unsigned gb = lid / tiles, trem = lid - gb * tiles;
  const unsigned grp = gb / (unsigned)p.batch, b = gb - grp * (unsigned)p.batch;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = p.K / 64;
  const half_t* A = p.dAp ? p.dAp[gb] : p.A[grp] + (size_t)b * p.sA;  // (device pointer tables: the dense twin behind sm_gemm_batched_*)
  const half_t* B = p.dBp ? p.dBp[gb] : p.B[grp] + (size_t)b * p.sB;
  half_t* C = p.dCp ? p.dCp[gb] : p.C[grp] + (size_t)b * p.sC;
  const int mlast = p.Mrows - 1;

  const char* asrc[SLA];
  const char* bsrc[SLB];
  unsigned aoff[SLA], boff[SLB];
#pragma unroll
  for (int i = 0; i < SLA; ++i) {  // 8 rows x 128 B: lane -> row 8t + lane/8, LDS chunk lane%8 holds source chunk (lane%8) ^ (row&7)
    const unsigned t = wave + (unsigned)NW * i;
    const unsigned row = 8u * t + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
    int gr = m0 + (int)row;
    gr = gr < mlast ? gr : mlast;
    asrc[i] = reinterpret_cast<const char*>(A + (size_t)gr * p.lda) + 16u * cs;
    aoff[i] = t * 1024u;
  }
#pragma unroll
  for (int i = 0; i < SLB; ++i) {
    const unsigned j = wave + (unsigned)NW * i, panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
    const unsigned cs = (lane & 7u) ^ b_swz(kr);
    int gc = n0 + (int)(64u * panel + 8u * cs);
    gc = gc <= p.N - 8 ? gc : p.N - 8;
    bsrc[i] = reinterpret_cast<const char*>(B + (size_t)kr * p.N + gc);
    boff[i] = BRING + panel * 8192u + (j & 7u) * 1024u;
  }
  const size_t bstep = (size_t)64 * p.N * 2;
  auto stage_a = [&](int kt, int buf) {  // A is read once by the whole grid when there is one column tile: non-temporal
#pragma unroll
    for (int i = 0; i < SLA; ++i) {
      if (ANT) __builtin_amdgcn_global_load_lds((gptr_t*)(asrc[i] + (size_t)kt * 128), (lptr_t*)(smem + buf * SA + aoff[i]), 16, 0, 2);
      else __builtin_amdgcn_global_load_lds((gptr_t*)(asrc[i] + (size_t)kt * 128), (lptr_t*)(smem + buf * SA + aoff[i]), 16, 0, 0);
    }
  };
  auto stage_b = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < SLB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t*)(bsrc[i] + (size_t)kt * bstep), (lptr_t*)(smem + buf * SB + boff[i]), 16, 0, 0);
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  // prologue = iterations -(NSA-1) .. -1 of the loop's issue pattern
#pragma unroll
  for (int j = -(NSA - 1); j < 0; ++j) {
    if (j + NSB - 1 >= 0 && j + NSB - 1 < nkt) stage_b(j + NSB - 1, j + NSB - 1);
    if (j + NSA - 1 < nkt) stage_a(j + NSA - 1, j + NSA - 1);
  }
  int ca = 0, cb = 0, fa = NSA - 1, fb = NSB - 1;  // current / next-to-fill slots of the two rings
  SM_T(unsigned long long tv = 0, tb = 0, ti = 0, ts = 0, tc = 0; unsigned long long s0 = sm_stamp(); const unsigned long long sstart = s0;)
  for (int kt = 0; kt < nkt; ++kt) {
#ifdef SM_STAMP
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AHEAD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long sv = sm_stamp(); tv += sv - s0;
    asm volatile("s_barrier" ::: "memory");
    const unsigned long long sb = sm_stamp(); tb += sb - sv;
#else
    if (kt + 1 < nkt) wait_dma_and_barrier<AHEAD>();
    else wait_dma_and_barrier<0>();
#endif
    if (kt + NSB - 1 < nkt) stage_b(kt + NSB - 1, fb);  // the slots stage kt - 1 occupied: every wave left them before this barrier
    if (kt + NSA - 1 < nkt) stage_a(kt + NSA - 1, fa);
#ifdef SM_STAMP
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long si = sm_stamp(); ti += si - sb;
    {
      const unsigned g = lane >> 4, r = lane & 15u;
      const char* Araw = smem + ca * SA;
      h8 af[FM];
      int idx[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const unsigned row = wave * TM + i * 16 + r;
        const u4 lo = *reinterpret_cast<const u4*>(Araw + a_off(row, 2u * g));
        const u4 hi = *reinterpret_cast<const u4*>(Araw + a_off(row, 2u * g + 1u));
        dense16_to_operand(lo, hi, af[i], idx[i]);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(af[i]), "+v"(idx[i]));
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long ss = sm_stamp(); ts += ss - si;
      smfmac_b_sweep<FM, FN, BF>(af, idx, smem + BRING + cb * SB, 0, lane, acc);
      __builtin_amdgcn_sched_barrier(0);
      s0 = sm_stamp(); tc += s0 - ss;
    }
#else
    if constexpr (DENSE) mfma_stage_dense_a<FM, FN, BF>(smem + ca * SA, smem + BRING + cb * SB, wave * TM, 0, lane, acc);
    else smfmac_stage_dense_a<FM, FN, BF>(smem + ca * SA, smem + BRING + cb * SB, wave * TM, 0, lane, acc);
#endif
    ca = ca + 1 == NSA ? 0 : ca + 1;
    fa = fa + 1 == NSA ? 0 : fa + 1;
    cb = cb + 1 == NSB ? 0 : cb + 1;
    fb = fb + 1 == NSB ? 0 : fb + 1;
  }
  __syncthreads();
  SM_T(const unsigned long long sloop = sm_stamp();)
  store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, true, wave * TM, 0, m0, n0, p.Mrows, p.N, p.alpha, p.beta, tid);
  SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8; const unsigned long long se = sm_stamp();
        d[0] = tv; d[1] = tb; d[2] = ti; d[3] = ts; d[4] = tc; d[5] = sloop - sstart; d[6] = se - sloop; })
}

template <int BN, bool BF = false, int NSA = 3, int NSB = 2, bool ANT = true, bool DENSE = false>
static int launch_fused_big(const FusedArgs& a0, hipStream_t st) {
  constexpr int BM = 256;
  FusedArgs a = a0;
#ifdef SM_TUNING
  if (!a.ablate) a.ablate = tuning_int("SM_DIRECT_ABLATE", 0) & 24;
#endif
  a.tiles_m = (a.Mrows + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch * a.ngroup;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_spmma_fused_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = (size_t)NSA * BM * 128 + (size_t)NSB * 64 * BN * 2;
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget of the big direct kernel");
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_fused_big_kernel<BN, BF, NSA, NSB, ANT, DENSE>), lds, "spmma_f16_fused_big_kernel")) return rc;
#ifdef SM_STAMP
  {
    static unsigned long long* dbg = nullptr;
    static size_t cap = 0;
    const size_t cnt = nwg * 8 * 8;
    if (cnt > cap) { if (dbg) (void)hipFree(dbg); (void)hipMalloc((void**)&dbg, cnt * 8); cap = cnt; }
    (void)hipMemset(dbg, 0, cnt * 8);
    a.dbg = dbg;
    spmma_f16_fused_big_kernel<BN, BF, NSA, NSB, ANT, DENSE><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(a);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(cnt);
    (void)hipMemcpy(h.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
    double t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < cnt / 8; ++i)
      for (int j = 0; j < 7; ++j) t[j] += (double)h[i * 8 + j];
    const double nwv = (double)(cnt / 8), nk = (double)(a.K / 64);
    fprintf(stderr, "STAMP-FUSED-BIG %dx%dx%d BN=%d NSA=%d NSB=%d tiles=%zu nkt=%d | per wave per stage: vmcnt-wait %.0f barrier %.0f dma-issue %.0f select %.0f b-sweep %.0f | loop %.0f epilogue %.0f cycles per tile\n",
            a.Mrows, a.N, a.K, BN, NSA, NSB, nwg, a.K / 64, t[0] / nwv / nk, t[1] / nwv / nk, t[2] / nwv / nk, t[3] / nwv / nk, t[4] / nwv / nk, t[5] / nwv, t[6] / nwv);
    return check_launch("spmma_f16_fused_big_kernel");
  }
#endif
  spmma_f16_fused_big_kernel<BN, BF, NSA, NSB, ANT, DENSE><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(a);
  return check_launch("spmma_f16_fused_big_kernel");
}

// a pointer the compiler cannot prove wave-uniform (an entry of the by-value pointer tables picked by a runtime index), made so
template <class T>
__device__ __forceinline__ T* sk_uniform(T* ptr) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
// (Round 5: a SOFTWARE-PIPELINED form of the big kernel -- the selection of stage kt + 1 interleaved, one strip behind every second B
//  fragment's SMFMACs, into the B sweep of stage kt; every wave loading exactly the rows it owns, so that A needed no barrier -- was
//  built, bit-identical, and measured 3-6 % SLOWER than the big kernel on all eight n > 128 shapes of the table
//  (profiles/ab_big2_r05k.txt): during the sweep the two waves of a SIMD already keep its matrix pipe issuing back to back, and vector
//  instructions slipped in between only delay the SMFMACs behind them.  Removed again: git history)

// (Round 5, second experiment: a SPLIT-ROLE form with a variable tile height -- the first NA waves issue only A pieces, the others only B,
//  so that an A wave's in-order vmcnt holds nothing but A and NSA - 2 whole stages stay in flight behind the one it waits for; tile
//  heights 256 / 224 / 192 / 160 / 128 rows (8 .. 4 waves of 32 rows, A rings of 3 .. 5 stages), so that 196 x 512 x 4608 x 3 has 150 /
//  168 / 198 / 240 / 294 tiles for the 256 CUs -- bit-identical, and within 0-5 % of the big kernel where it is dispatched
//  (196 x 512 x 4608 x 3: 114.0 / 110.1 / 109.1 / 110.7 / 163.8 us against 114.5; profiles/ab_big3_r05l.txt).  What that says: a tile's
//  stage takes the same ~3 700 cycles at 160 rows as at 256 -- every wave owns 32 rows and ALL 256 columns whatever the tile's
//  height, so the stage is one wave's own serial chain (its 8-11 DMA pieces, its 128 selection instructions at half a SIMD's issue
//  rate, its 32 SMFMACs), not the A stream's latency and not the tile's bytes.  Removed again: git history)

// ---------------------------------------------------------------------------------------------
// STREAM-K form of the big kernel (round 5): the shapes with few row tiles and a long K (196 x 512 x 4608: 150 tiles of 72 stages on
// 256 CUs; 784 x 512 x 1024: 196 tiles) leave the last -- often the only -- round of one-per-CU workgroups 23-43 % empty, and a tile's
// time is its K stages.  Here the launch's work is the list of STAGE UNITS (row panel major, stage inside the panel), cut into
// contiguous ranges ("slots", SkPlan below: groups of tg panels cut into wg equal ranges), one per workgroup and column tile, at most
// one workgroup per CU: a workgroup walks its range segment by segment (a segment = a run of stages of one tile), each segment
// through the big kernel's own pipeline.  A tile that lies whole inside one range is computed
// and stored exactly as the big kernel does (bit-identical); a tile cut by range borders is summed from fp32 partials in a FIXED
// order -- no atomics on data, repeated runs give the same bits:
//   * the workgroup that holds the tile's FIRST stages (they are the LAST segment of its range) owns the tile: it finishes its
//     segment, then adds the partials of the same column tile's workgroups of the slots after it (ascending k) to its
//     accumulators and runs the ordinary epilogue (alpha / beta, one rounding);
//   * the other holders meet the tile in the FIRST segment of their range: they store their accumulators to their slot of the
//     workspace (256 x BN fp32, register-image order: 1 KiB per wave instruction, no transposition on either side) and raise
//     their flag -- long before the owner, whose range ends where theirs begins, comes looking for it.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, third table row): partials leave by `global_store_dwordx4 sc1` (write-
// through, whole 128-byte lines per wave instruction), every storing wave drains vmcnt, a workgroup barrier, ONE lane's agent-scope
// atomic add on the slot's flag; the owner's lane 0 polls the flag with agent-scope relaxed loads, a workgroup barrier, then every
// wave reads with `global_load_dwordx4 sc1`.  The owner zeroes the flag again once every wave has read (the next launch on the
// stream finds the workspace as this one did: zero flags -- hipGraph replays need no memset node).
// Progress: an owner only waits for workgroups of HIGHER logical id, which produce what it waits for first thing after they start,
// and the grid never exceeds the CU count (one workgroup per CU: the whole LDS), so they are resident or next in the dispatch order.
// ---------------------------------------------------------------------------------------------
struct SkArgs {
  float* part;             // one slot of 256 x BN fp32 partial sums per workgroup
  unsigned* flags;         // one word per workgroup; zero before the launch, zero after it
  // the decomposition: row panels (256 rows of one problem: tiles_m x batch x ngroup of them) x nkt stage units; `slots` workgroup
  // slots, each a contiguous range of panel units, every slot run by tiles_n workgroups in lockstep (one per column tile);
  // a GROUP = tg consecutive panels cut into wg slot ranges at cut[0 .. wg] (units from the group's first; cut[0] = 0, cut[wg] =
  // tg * nkt); the last group: tgl panels, wgl slots, cutl[]
  unsigned panels, slots, tg, wg, groups_full, tgl, wgl;
  unsigned cut[9], cutl[9];
#ifdef SM_TUNING
  int ablate;              // tuning builds only: 1 = no partial stores, 2 = no fix-up loads / waits (results wrong; timing only)
#endif
};
// entry r of a 9-entry table held in the kernel arguments, by compares (a runtime index into a by-value argument array would
// send the whole block through scratch memory)
__device__ __forceinline__ unsigned sk_pick(const unsigned (&tab)[9], unsigned r) {
  unsigned v = tab[0];
#pragma unroll
  for (unsigned i = 1; i < 9; ++i) v = r == i ? tab[i] : v;
  return v;
}
// first panel unit of slot w (w == slots: the end of the launch's units)
__device__ __forceinline__ unsigned long long sk_slot_start(const SkArgs& s, unsigned w, unsigned nkt) {
  const unsigned q = w / s.wg;
  if (q < s.groups_full) return ((unsigned long long)q * s.tg) * nkt + sk_pick(s.cut, w - q * s.wg);
  unsigned r = w - s.groups_full * s.wg;
  r = r < s.wgl ? r : s.wgl;
  return ((unsigned long long)s.groups_full * s.tg) * nkt + (s.wgl ? sk_pick(s.cutl, r) : 0u);
}
__device__ __forceinline__ void sk_store_sc1(float* dst, f4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ f4 sk_load_sc1(const float* src) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(src) : "memory");
  return v;
}

template <int BN, bool BF = false, bool ANT = true, bool DENSE = false>
__global__ __launch_bounds__(512) void spmma_f16_fused_sk_kernel(const FusedArgs p, const SkArgs s) {
  constexpr int BM = 256, NW = 8, TM = BM / NW, FM = TM / 16, FN = BN / 16, NSA = 3, NSB = 2;
  constexpr int SA = BM * 128, SB = 64 * BN * 2;
  constexpr int A_N = BM / 8, B_N = BN / 8;  // 1 KiB DMA wave-instructions per stage
  static_assert(A_N % NW == 0 && B_N % NW == 0, "equal DMA share per wave");
  constexpr int SLA = A_N / NW, SLB = B_N / NW;
  constexpr int AHEAD = SLA;  // pieces issued after B(kt) that may stay in flight at iteration kt: A(kt + 1)
  constexpr int BRING = NSA * SA;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SM_TUNING
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, (p.ablate & 8) != 0);  // (tuning: SM_DIRECT_ABLATE bit 3 = dispatch order; measured 0-5 % slower here, mma_tile.h)
#else
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
#endif
  const unsigned tn = (unsigned)p.tiles_n;
  const unsigned slot = lid / tn, tile_n = lid - slot * tn;  // the tiles_n workgroups of a slot are neighbours: one XCD, one A stream
  const unsigned nkt = (unsigned)(p.K / 64);
  const unsigned long long u0 = sk_slot_start(s, slot, nkt), u1 = sk_slot_start(s, slot + 1u, nkt);
  const int mlast = p.Mrows - 1;
  const size_t bstep = (size_t)64 * p.N * 2;
  // this lane's piece of a partial slot: fragment (i, j) of wave w at ((w * FM * FN + i * FN + j) * 64 + lane) float4s
  const size_t slot_floats = (size_t)BM * BN;

  bool publish_pending = false;  // this workgroup's partial stores are issued, its flag not yet raised
  for (unsigned long long u = u0; u < u1;) {
    const unsigned t = (unsigned)(u / nkt);
    const unsigned kb = (unsigned)(u - (unsigned long long)t * nkt);
    unsigned len = nkt - kb;
    if (u1 - u < (unsigned long long)len) len = (unsigned)(u1 - u);
    const unsigned ke = kb + len;
    const unsigned gb = t / (unsigned)p.tiles_m, tile_m = t - gb * (unsigned)p.tiles_m;  // t: the row panel
    const unsigned grp = gb / (unsigned)p.batch, b = gb - grp * (unsigned)p.batch;
    const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
    const half_t* A = sk_uniform(p.dAp ? p.dAp[gb] : p.A[grp] + (size_t)b * p.sA);
    const half_t* B = sk_uniform(p.dBp ? p.dBp[gb] : p.B[grp] + (size_t)b * p.sB);
    half_t* C = sk_uniform(p.dCp ? p.dCp[gb] : p.C[grp] + (size_t)b * p.sC);

    // (the segment loop's per-lane values that do not depend on the tile -- DMA chunk swizzles, epilogue addresses, slot offsets --
    // are recomputed per segment from opaque copies of the lane / thread id: hoisted out of the loop they would live across the
    // main loop, whose 128 accumulators leave them no registers, and spill)
    unsigned lane_p = lane;
    asm volatile("" : "+v"(lane_p));
    // per-lane DMA sources as 32-bit offsets from two uniform bases (the launcher checks that a tile's rows and a stage of B stay
    // below 4 GiB): the accumulators and the segment loop leave no room for 64-bit addresses
    const char* const abase = reinterpret_cast<const char*>(A + (size_t)m0 * p.lda) + (size_t)kb * 128;
    const char* const bbase = reinterpret_cast<const char*>(B) + (size_t)kb * bstep;
    unsigned asrc[SLA], bsrc[SLB], aoff[SLA], boff[SLB];
#pragma unroll
    for (int i = 0; i < SLA; ++i) {  // 8 rows x 128 B: lane -> row 8t + lane/8, LDS chunk lane%8 holds source chunk (lane%8) ^ (row&7)
      const unsigned tt = wave + (unsigned)NW * i;
      const unsigned row = 8u * tt + (lane_p >> 3), cs = (lane_p & 7u) ^ (row & 7u);
      int lr = (int)row;
      lr = m0 + lr < mlast ? lr : mlast - m0;
      asrc[i] = (unsigned)lr * (unsigned)p.lda * 2u + 16u * cs;
      aoff[i] = tt * 1024u;
    }
#pragma unroll
    for (int i = 0; i < SLB; ++i) {
      const unsigned j = wave + (unsigned)NW * i, panel = j >> 3, kr = 8u * (j & 7u) + (lane_p >> 3);
      const unsigned cs = (lane_p & 7u) ^ b_swz(kr);
      int gc = n0 + (int)(64u * panel + 8u * cs);
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      bsrc[i] = (kr * (unsigned)p.N + (unsigned)gc) * 2u;
      boff[i] = BRING + panel * 8192u + (j & 7u) * 1024u;
    }
    // buffer-addressed LDS-DMA: 32-bit per-lane offsets + a scalar stage offset against a uniform resource (no 64-bit address
    // arithmetic per piece, half the address registers of the global form)
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(abase), 0, (int)0xffffffffu, 0x27000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(bbase), 0, (int)0xffffffffu, 0x27000);
    auto stage_a = [&](int kt, int buf) {  // kt counts from the segment's first stage
#pragma unroll
      for (int i = 0; i < SLA; ++i) {
        if (ANT) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lptr_t*)(smem + buf * SA + aoff[i]), 16, (int)asrc[i], kt * 128, 0, 2);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lptr_t*)(smem + buf * SA + aoff[i]), 16, (int)asrc[i], kt * 128, 0, 0);
      }
    };
    auto stage_b = [&](int kt, int buf) {
#pragma unroll
      for (int i = 0; i < SLB; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lptr_t*)(smem + buf * SB + boff[i]), 16, (int)bsrc[i], kt * (int)bstep, 0, 0);
    };

    f4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int nst = (int)len;
    // prologue = iterations -(NSA-1) .. -1 of the loop's issue pattern (the big kernel's)
#pragma unroll
    for (int j = -(NSA - 1); j < 0; ++j) {
      if (j + NSB - 1 >= 0 && j + NSB - 1 < nst) stage_b(j + NSB - 1, j + NSB - 1);
      if (j + NSA - 1 < nst) stage_a(j + NSA - 1, j + NSA - 1);
    }
    int ca = 0, cb = 0, fa = NSA - 1, fb = NSB - 1;
    for (int kt = 0; kt < nst; ++kt) {
      if (kt + 1 < nst) wait_dma_and_barrier<AHEAD>();
      else wait_dma_and_barrier<0>();
      if (publish_pending) {
        // the previous segment's partial stores are older than everything this segment issued: the counted wait above covered them
        // for this wave, the barrier for all eight -- their drain was hidden behind this segment's first loads
        if (tid == 0) (void)__hip_atomic_fetch_add(s.flags + lid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        publish_pending = false;
      }
      if (kt + NSB - 1 < nst) stage_b(kt + NSB - 1, fb);
      if (kt + NSA - 1 < nst) stage_a(kt + NSA - 1, fa);
      if constexpr (DENSE) mfma_stage_dense_a<FM, FN, BF>(smem + ca * SA, smem + BRING + cb * SB, wave * TM, 0, lane, acc);
      else smfmac_stage_dense_a<FM, FN, BF>(smem + ca * SA, smem + BRING + cb * SB, wave * TM, 0, lane, acc);
      ca = ca + 1 == NSA ? 0 : ca + 1;
      fa = fa + 1 == NSA ? 0 : fa + 1;
      cb = cb + 1 == NSB ? 0 : cb + 1;
      fb = fb + 1 == NSB ? 0 : fb + 1;
    }
    __syncthreads();  // every wave has left the rings; nothing is in flight (the last iteration waited for vmcnt(0))
    unsigned tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const size_t lane_off = ((size_t)wave * (FM * FN) * 64u + (tid_e & 63u)) * 4u;

    if (kb != 0) {
      // ---- a later part of a tile that another workgroup owns: publish the fp32 partial sums, raise the flag
      float* dst = s.part + (size_t)lid * slot_floats + lane_off;
      asm volatile("" : "+v"(dst));  // (keeps the 32 store addresses from being computed ahead of the segment loop and spilled)
      bool publish = true;
#ifdef SM_TUNING
      publish = !(s.ablate & 1);
#endif
      if (publish) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) sk_store_sc1(dst + (size_t)(i * FN + j) * 256u, acc[i][j]);
      }
      if (u + len < u1) {
        publish_pending = true;  // another segment follows: it raises the flag behind its first stage's wait (above)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) (void)__hip_atomic_fetch_add(s.flags + lid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (ke != nkt) {
        // ---- the tile's first stages, and others hold the rest: add their partials in ascending k (= ascending logical id)
        const unsigned long long tile_end = (unsigned long long)(t + 1u) * nkt;
        unsigned c_end = s.slots;
#ifdef SM_TUNING
        if (s.ablate & 2) c_end = 0;
#endif
        for (unsigned cs = slot + 1u; cs < c_end && sk_slot_start(s, cs, nkt) < tile_end; ++cs) {
          const unsigned c = cs * tn + tile_n;  // the workgroup of slot cs that holds this column tile
          if (tid == 0) {
            // (bounded: ~0.3 s of polling -- five orders of magnitude beyond any legitimate wait -- then word 1023 of the flag page is
            //  raised and the tile is stored without the missing partial: a broken hand-off must show as a wrong result and a
            //  non-zero flag page, never as a hung device)
            unsigned spins = 0;
            while (__hip_atomic_load(s.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(2);
              if (++spins > (1u << 22)) {
                __hip_atomic_store(s.flags + 1023, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
          }
          __syncthreads();
          // (buffer loads with the sc1 bit: the compiler counts their vmcnt itself, so the 32 fragments need no hand-made waits)
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(s.part + (size_t)c * slot_floats, 0, (int)(slot_floats * 4u), 0x27000);
          const unsigned voff = (unsigned)lane_off * 4u;
#pragma unroll
          for (int f0 = 0; f0 < FM * FN; f0 += 8) {
            u4 v[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) v[f] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (unsigned)(f0 + f) * 1024u, 0, 16 /* sc1 */);
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[(f0 + f) / FN][(f0 + f) % FN] += __builtin_bit_cast(f4, v[f]);
          }
          __syncthreads();  // every wave has read slot c: hand the flag back as it was found
          if (tid == 0) __hip_atomic_store(s.flags + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, true, wave * TM, 0, m0, n0, p.Mrows, p.N, p.alpha, p.beta, tid_e);
      __syncthreads();  // the epilogue image aliases the rings the next segment's DMA fills
    }
    u += len;
  }
}

// bytes of the stream-K workspace for a launch of `nwg` workgroups (flags first, then the slots)
static size_t sk_workspace_bytes(size_t nwg, int bn) { return 4096 + nwg * 256 * (size_t)bn * 4; }

// The decomposition of a launch (host side): `panels` row panels of nkt stage units, tn column tiles, at most cus / tn slots.
// A group of tg consecutive panels is cut into wg equal slot ranges; all groups have the same cuts, so at any moment the chip's
// workgroups stand at only wg different K positions (and the tn workgroups of a slot at the same one): the B stage a workgroup
// brings in is the one its neighbours on the XCD's L2 want now, and the two column tiles of a row panel stream A once.  (The first
// form of this kernel cut the tile-major unit list into one range per CU, 256 different K positions: bit-correct, but B and the
// second column tile's A then missed L2 and the launch ran at the fabric's rate -- 196 x 512 x 4608 x 3: 106 us for the main loops
// alone against 126 us for whole tiles on 150 CUs, profiles/sk_probe_r05b.txt.)  Chosen: the (tg, wg <= 8) with the fewest units
// per slot, ties to the smaller wg.
struct SkPlan {
  unsigned tg, wg, groups_full, tgl, wgl, slots, units;  // units: the longest slot range
  unsigned cut[9], cutl[9];
};
// Where the wg ranges of a group of `tiles` panels are cut.  Equal ranges would make every workgroup finish at the same moment -- but
// a range that ENDS inside a tile it does not own (its last segment starts past the tile's first stage) ends with a publish, and
// the owner of that tile, finishing its own range at that same moment, then waits for the publish and the hand-off before it can
// read the partial (measured: ~14 us at the end of a 100 us launch).  Such ranges are made `lead` units shorter, the others take up
// the difference: the partial is in memory when the owner comes for it.  Returns the longest range.
static unsigned sk_cuts(unsigned tiles, unsigned wg, unsigned nkt, unsigned lead, unsigned (&cut)[9]) {
  const double total = (double)tiles * nkt;
  double x[8];
  for (unsigned r = 0; r < wg; ++r) x[r] = total / wg;
  for (int it = 0; it < 6; ++it) {
    double acc = 0;
    cut[0] = 0;
    for (unsigned r = 0; r < wg; ++r) {
      acc += x[r];
      cut[r + 1] = r + 1 == wg ? (unsigned)total : (unsigned)(acc + 0.5);
    }
    unsigned nshort = 0;
    bool shrt[8];
    for (unsigned r = 0; r < wg; ++r) {
      const unsigned last_tile = (cut[r + 1] - 1) / nkt;                      // the tile the range ends in
      const unsigned seg0 = cut[r] > last_tile * nkt ? cut[r] : last_tile * nkt;  // where its last segment starts
      shrt[r] = cut[r + 1] > cut[r] && seg0 != last_tile * nkt;               // not at the tile's first stage: ends as a contributor
      nshort += shrt[r] ? 1u : 0u;
    }
    const double T = (total + (double)lead * nshort) / wg;
    if (T - lead < 2.0) break;  // (ranges too short to take a lead from: equal cuts)
    for (unsigned r = 0; r < wg; ++r) x[r] = shrt[r] ? T - lead : T;
  }
  unsigned longest = 0;
  for (unsigned r = 0; r < wg; ++r) {
    if (cut[r + 1] < cut[r]) cut[r + 1] = cut[r];
    longest = cut[r + 1] - cut[r] > longest ? cut[r + 1] - cut[r] : longest;
  }
  for (unsigned r = wg + 1; r < 9; ++r) cut[r] = cut[wg];
  return longest;
}
static SkPlan sk_plan_for(size_t panels, size_t nkt, unsigned tg, unsigned wg, unsigned lead) {
  SkPlan pl = {};
  pl.tg = tg; pl.wg = wg;
  pl.groups_full = (unsigned)(panels / tg);
  pl.tgl = (unsigned)(panels % tg);
  pl.wgl = pl.tgl ? (pl.tgl * wg + tg - 1) / tg : 0;
  pl.slots = pl.groups_full * wg + pl.wgl;
  pl.units = pl.groups_full ? sk_cuts(tg, wg, (unsigned)nkt, lead, pl.cut) : 0;
  if (pl.tgl) {
    const unsigned ul = sk_cuts(pl.tgl, pl.wgl, (unsigned)nkt, lead, pl.cutl);
    pl.units = ul > pl.units ? ul : pl.units;
  }
  return pl;
}
constexpr unsigned SK_LEAD = 3;  // stage units (~5 us) a publishing range ends ahead of the owner that reads it
static SkPlan sk_plan(size_t panels, size_t nkt, size_t tn, size_t cus) {
  SkPlan best = {};
  const size_t max_slots = cus / tn;
  if (max_slots == 0 || panels == 0 || nkt == 0 || nkt > 0xffffu) return best;
  const unsigned lead = (unsigned)tuning_int("SM_SK_LEAD", (int)SK_LEAD);
  for (unsigned wg = 1; wg <= 8; ++wg)
    for (unsigned tg = 1; tg <= 16; ++tg) {
      if (wg > 1 && (size_t)tg * nkt < 2 * (size_t)wg) continue;  // at least two stages per range
      const size_t gf = panels / tg, tgl = panels % tg, wgl = tgl ? (tgl * wg + tg - 1) / tg : 0;
      if (gf * wg + wgl == 0 || gf * wg + wgl > max_slots) continue;
      const SkPlan pl = sk_plan_for(panels, nkt, tg, wg, lead);
      if (best.slots == 0 || pl.units < best.units) best = pl;
    }
  return best;
}

template <int BN, bool BF = false, bool ANT = true, bool DENSE = false>
static int launch_fused_sk(const FusedArgs& a0, void* workspace, size_t workspace_bytes, hipStream_t st) {
  constexpr int BM = 256;
  FusedArgs a = a0;
#ifdef SM_TUNING
  if (!a.ablate) a.ablate = tuning_int("SM_DIRECT_ABLATE", 0) & 24;
#endif
  a.tiles_m = (a.Mrows + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t panels = (size_t)a.tiles_m * a.batch * a.ngroup;
  if (panels == 0) return SM_STATUS_SUCCESS;
  const size_t nkt = (size_t)a.K / 64;
  if (panels > 0x7fffffu || nkt == 0 || nkt > 0xffffu) {
    set_error("sm_spmma_fused_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  if ((size_t)256 * a.lda * 2 + (size_t)a.K * 2 >= ((size_t)1 << 32) || (size_t)(a.K + 64) * a.N * 2 >= ((size_t)1 << 31)) {
    set_error("sm_spmma_fused_*_ws: leading dimension too large for the stream-K form's 32-bit tile offsets");
    return SM_STATUS_NOT_SUPPORTED;
  }
  const int cus = device_cu_count();
  SkPlan pl = sk_plan(panels, nkt, (size_t)a.tiles_n, (size_t)cus);
#ifdef SM_TUNING
  if (const int wg_env = tuning_int("SM_SK_WG", 0)) {  // tuning: force the group shape (SM_SK_TG panels cut into SM_SK_WG ranges)
    const unsigned tg = (unsigned)tuning_int("SM_SK_TG", 1), wg = (unsigned)(wg_env > 8 ? 8 : wg_env);
    pl = sk_plan_for(panels, nkt, tg ? tg : 1, wg, (unsigned)tuning_int("SM_SK_LEAD", (int)SK_LEAD));
  }
#endif
  const size_t nwg = (size_t)pl.slots * a.tiles_n;
  if (pl.slots == 0 || nwg > (size_t)cus || nwg > 1023) {
    set_error("sm_spmma_fused_*_ws: the shape has more column tiles than the stream-K form has compute units for");
    return SM_STATUS_NOT_SUPPORTED;
  }
  if (!workspace || !aligned16(workspace) || workspace_bytes < sk_workspace_bytes(nwg, BN)) {
    set_error("sm_spmma_fused_*_ws: the stream-K form needs a 16-byte aligned workspace of sm_spmma_fused_workspace_size bytes");
    return SM_STATUS_INVALID_VALUE;
  }
  SkArgs s = {};
  s.flags = reinterpret_cast<unsigned*>(workspace);
  s.part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + 4096);
  s.panels = (unsigned)panels; s.slots = pl.slots; s.tg = pl.tg; s.wg = pl.wg; s.groups_full = pl.groups_full; s.tgl = pl.tgl; s.wgl = pl.wgl;
  for (int i = 0; i < 9; ++i) { s.cut[i] = pl.cut[i]; s.cutl[i] = pl.cutl[i]; }
#ifdef SM_TUNING
  s.ablate = tuning_int("SM_SK_ABLATE", 0);
#endif
  constexpr size_t lds_main = (size_t)3 * BM * 128 + (size_t)2 * 64 * BN * 2;
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS budget of the stream-K kernel");
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_fused_sk_kernel<BN, BF, ANT, DENSE>), lds, "spmma_f16_fused_sk_kernel")) return rc;
  spmma_f16_fused_sk_kernel<BN, BF, ANT, DENSE><<<dim3((unsigned)nwg), dim3(512), lds, st>>>(a, s);
  return check_launch("spmma_f16_fused_sk_kernel");
}

// The dispatch rule of the workspace entry points: the stream-K form where the rounds of whole tiles leave CUs idle.  Costs in
// stages of a 256 x 256 tile: stream-K = its longest slot range + ~8 stages for the fix-up (measured: 12-16 us -- every owner reads
// its one or two 256 KiB partial tiles at the end of the launch, 40-100 MB at the chip's memory rate all at once: profiles/
// sk_probe_r05d.txt); the big form = rounds x stages; a 128 x 256 tile's stage (wide kernel) takes ~0.6 of a 256 x 256 one
// (profiles/stamp_r04b_big_wide_direct.txt).  The A-stationary shapes (n > 256, k <= 512, plain store) keep their kernel: few
// stages per tile, nothing for a K split to balance.  SM_FUSED_SK (tuning): 0 = never, 2 = wherever the kernel takes the shape.
static bool sk_takes(size_t rows, size_t problems, size_t n, size_t k, bool astat_shape, SkPlan& pl) {
  const int sk_rule = tuning_int("SM_FUSED_SK", 1);
  if (!sk_rule || n <= 128 || k < 128 || k % 64 != 0) return false;
  const size_t cus = (size_t)device_cu_count(), nkt = k / 64;
  const size_t panels = (rows + 255) / 256 * problems, tn = (n + 255) / 256;
  const size_t t_big = panels * tn, t_wide = (rows + 127) / 128 * tn * problems;
  pl = sk_plan(panels, nkt, tn, cus);
  if (!pl.slots) return false;
  if (sk_rule == 2) return true;
  const double c_sk = (double)pl.units + (pl.wg > 1 ? 8.0 : 0.0);
  const double c_big = (double)((t_big + cus - 1) / cus * nkt), c_wide = 0.6 * (double)((t_wide + cus - 1) / cus * nkt);
  const double c_now = c_big < c_wide ? c_big : c_wide;
  return !astat_shape && pl.wg > 1 && c_sk < 0.85 * c_now;
}

// ---------------------------------------------------------------------------------------------
// Rows that are NOT whole 64-deep stages of 16-byte aligned pieces (k % 64 != 0 or k % 8 != 0: the 7 x 7 x 3 stem layer
// of every ResNet, k = 147), n <= 128, one tall contiguous A (lda == k, batches back to back): SPAN form.  A tile's 128
// rows are ONE contiguous span of 128 * k * 2 bytes that starts on a 256-byte boundary, so it reaches LDS by plain
// 1 KiB LDS-DMA pieces whatever the row pitch (per-lane source addresses clamped to the operand's last 16 bytes); the
// whole B (round_up(k, 64) rows, rows at or beyond k from a zero page: 0 x inf must not poison finite outputs) arrives
// with it: one wait, one barrier, then every stage is computed out of LDS -- the lane that feeds (row, k-group) to the
// SMFMAC picks its 16 dense halves with 2-byte LDS reads at row * k * 2 + ..., zeroes what lies at or beyond k (the
// virtual zeros that complete a ragged strip, oracle: strip_select) and selects as the direct kernel does.  Same operands,
// same instruction sequence per stage as compress + spmma: bit-identical C.  Two workgroups per CU (61 KiB at k = 147,
// n = 64) overlap each other's load and compute phases.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) const unsigned char sm_fused_zero_page[256] = {0};

template <int BN, bool BF = false, bool DENSE = false>
__global__ __launch_bounds__(256) void spmma_f16_fused_span_kernel(const FusedArgs p, const unsigned span_lds /*bytes reserved for the A span*/,
                                                                   const size_t a_bytes /*bytes of one problem's A*/) {
  constexpr int BM = 128, NW = 4, TM = BM / NW, FM = TM / 16, FN = BN / 16;
  constexpr int SB = 64 * BN * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
#ifdef SM_TUNING
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, (p.ablate & 8) != 0);  // (tuning: SM_DIRECT_ABLATE bit 3 = dispatch order; measured 0-5 % slower here, mma_tile.h)
#else
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
#endif
  const unsigned grp = lid / tiles, trem = lid - grp * tiles;  // batch == 1: the batches are stacked rows
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = (p.K + 63) / 64;
  const char* A = reinterpret_cast<const char*>(p.dAp ? p.dAp[grp] : p.A[grp]);  // (device pointer tables: the dense twin behind sm_gemm_batched_*)
  const half_t* B = p.dBp ? p.dBp[grp] : p.B[grp];
  half_t* C = p.dCp ? p.dCp[grp] : p.C[grp];
  const unsigned rowbytes = (unsigned)p.K * 2u;
  const int rows = p.Mrows - m0 < BM ? p.Mrows - m0 : BM;  // valid rows of this tile (>= 1)

  // ---- A span: bytes [m0 * rowbytes, (m0 + rows) * rowbytes) of the operand, in 1 KiB pieces
  {
    const size_t s0 = (size_t)m0 * rowbytes;               // multiple of 256
    const unsigned len = (unsigned)rows * rowbytes;
    const unsigned np = (len + 1023u) / 1024u;
    const size_t last16 = a_bytes - 16;                     // a_bytes % 16 == 0 (launcher)
    for (unsigned pc = wave; pc < np; pc += NW) {
      size_t off = s0 + (size_t)pc * 1024u + 16u * lane;
      off = off < last16 ? off : last16;
      __builtin_amdgcn_global_load_lds((gptr_t*)(A + off), (lptr_t*)(smem + pc * 1024u), 16, 0, 2);
    }
  }
  // ---- B: nkt stage images [64][BN] (the direct kernel's layout); k rows at or beyond K come from the zero page
  char* const Bimg = smem + span_lds;
  {
    constexpr int B_N = BN / 8;  // pieces per stage
    const int npb = nkt * B_N;
    for (int t = (int)wave; t < npb; t += NW) {
      const int kt = t / B_N, j = t - kt * B_N;
      const unsigned panel = (unsigned)j >> 3, kr = 8u * ((unsigned)j & 7u) + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ b_swz(kr);
      int gc = n0 + (int)(64u * panel + 8u * cs);
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      const int krow = kt * 64 + (int)kr;
      const char* src = krow < p.K ? reinterpret_cast<const char*>(B + (size_t)krow * p.N + gc)
                                   : reinterpret_cast<const char*>(sm_fused_zero_page) + 16u * (lane & 7u);
      __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(Bimg + kt * SB + panel * 8192u + ((unsigned)j & 7u) * 1024u), 16, 0, 0);
    }
  }
  wait_dma_and_barrier<0>();

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned g = lane >> 4, r = lane & 15u;
  unsigned rowoff[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int row = (int)(wave * TM + i * 16 + r);
    row = row < rows ? row : rows - 1;  // rows past the edge re-read the last valid one (their outputs are never stored)
    rowoff[i] = (unsigned)row * rowbytes;
  }
  for (int kt = 0; kt < nkt; ++kt) {
    if constexpr (DENSE) {
      // the dense twin: the lane's 8 + 8 halves (k = 8 g .. + 7 of the stage's two 32-k blocks), those at or beyond k zeroed
      h8 a0[FM], a1[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        uint32_t d[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k0 = kt * 64 + 32 * h + 8 * (int)g;
          int nv = p.K - k0;
          nv = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
          const char* src = smem + rowoff[i] + 2u * (unsigned)k0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t lo16 = *reinterpret_cast<const unsigned short*>(src + 4 * e);
            const uint32_t hi16 = *reinterpret_cast<const unsigned short*>(src + 4 * e + 2);
            uint32_t v = lo16 | (hi16 << 16);
            v = 2 * e + 1 < nv ? v : (2 * e < nv ? (v & 0xffffu) : 0u);
            d[h][e] = v;
          }
        }
        a0[i] = __builtin_bit_cast(h8, u4{d[0][0], d[0][1], d[0][2], d[0][3]});
        a1[i] = __builtin_bit_cast(h8, u4{d[1][0], d[1][1], d[1][2], d[1][3]});
      }
      mfma_b_sweep<FM, FN, BF>(a0, a1, Bimg + kt * SB, 0, lane, acc);
      continue;
    }
    h8 af[FM];
    int idx[FM];
    const int k0 = kt * 64 + 16 * (int)g;
    int nv = p.K - k0;  // valid elements of this lane's 16
    nv = nv < 0 ? 0 : (nv > 16 ? 16 : nv);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const char* src = smem + rowoff[i] + 2u * (unsigned)k0;
      uint32_t d[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // (reads at or beyond k are masked below; they stay inside the LDS allocation: span_lds covers 128 rows + 128 bytes)
        const uint32_t lo16 = *reinterpret_cast<const unsigned short*>(src + 4 * e);
        const uint32_t hi16 = *reinterpret_cast<const unsigned short*>(src + 4 * e + 2);
        uint32_t v = lo16 | (hi16 << 16);
        v = 2 * e + 1 < nv ? v : (2 * e < nv ? (v & 0xffffu) : 0u);
        d[e] = v;
      }
      dense16_to_operand(u4{d[0], d[1], d[2], d[3]}, u4{d[4], d[5], d[6], d[7]}, af[i], idx[i]);
    }
    smfmac_b_sweep<FM, FN, BF>(af, idx, Bimg + kt * SB, 0, lane, acc);
  }
  __syncthreads();
  store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, true, wave * TM, 0, m0, n0, p.Mrows, p.N, p.alpha, p.beta, tid);
}

template <int BN, bool BF = false, bool DENSE = false>
static int launch_fused_span(const FusedArgs& a0, hipStream_t st) {
  FusedArgs a = a0;
#ifdef SM_TUNING
  if (!a.ablate) a.ablate = tuning_int("SM_DIRECT_ABLATE", 0) & 24;
#endif
  a.tiles_m = (a.Mrows + 127) / 128;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.ngroup;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  const size_t kc = ((size_t)a.K + 63) / 64 * 64;
  const size_t span_lds = ((size_t)128 * a.K * 2 + 128 + 1023) / 1024 * 1024;  // whole pieces; >= 128 bytes past the last row
  const size_t lds_main = span_lds + kc * BN * 2;
  constexpr size_t lds_epi = (size_t)128 * (BN * 2 + 16);
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  if (lds > 160 * 1024 || nwg > 0x7fffffffu) {
    set_error("sm_spmma_fused_{f16,bf16}: k too long for the span form (use the staged path)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_fused_span_kernel<BN, BF, DENSE>), 160 * 1024, "spmma_f16_fused_span_kernel")) return rc;
  spmma_f16_fused_span_kernel<BN, BF, DENSE><<<dim3((unsigned)nwg), dim3(256), lds, st>>>(a, (unsigned)span_lds, (size_t)a.Mrows * a.K * 2);
  return check_launch("spmma_f16_fused_span_kernel");
}

// The B tile's LDS-DMA, shared by the B loader waves of the wide kernel and -- when it has none (NLB = 0) -- by its
// consumer waves: instruction j = bw + NBW * i of a stage covers k-rows 8 * (j & 7) + lane / 8 of panel j >> 3.
template <int BN, int NBW>
struct BTileDma {
  static constexpr int B_WI = (BN / 8) / NBW;
  const char* src[B_WI];
  unsigned dst[B_WI];
  size_t step;
  __device__ __forceinline__ void setup(const half_t* B, int N, int n0, unsigned bw, unsigned lane, unsigned ring_off) {
#pragma unroll
    for (int i = 0; i < B_WI; ++i) {
      const unsigned j = bw + (unsigned)NBW * i, panel = j >> 3, grp = j & 7u, kr = 8u * grp + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ b_swz(kr);
      int gc = n0 + (int)(64u * panel + 8u * cs);
      gc = gc <= N - 8 ? gc : N - 8;
      src[i] = reinterpret_cast<const char*>(B) + ((size_t)kr * N + (size_t)gc) * 2;
      dst[i] = ring_off + panel * 8192u + grp * 1024u;
    }
    step = (size_t)64 * N * 2;
  }
  __device__ __forceinline__ void issue(char* smem, int kt, unsigned buf_off) const {
#pragma unroll
    for (int i = 0; i < B_WI; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)kt * step), (lptr_t*)(smem + buf_off + dst[i]), 16, 0, 0);
  }
};

// (A register-A form of the direct kernel -- each SMFMAC lane loading the 32 bytes of its row straight from global memory,
// PF stages ahead in registers, only B through LDS -- was built in round 2: bit-identical, 1.3x slower on every n <= 128 layer
// (profiles/tune_rega_r02q.txt; 164 VGPRs, half-line wave loads).  Removed again; DESIGN.md 4.5, git history.)

template <int BN, int WM, int WN, int NLB, int PF, int NSB, bool BF = false, bool ANT = true>
__global__ __launch_bounds__(64 * (WM * WN + 4 + NLB)) void spmma_f16_fused_wide_kernel(const FusedArgs p) {
  constexpr int BM = 128, NLA = 4, NC = WM * WN, NW = NC + NLA + NLB;
  static_assert(PF >= 1 && PF <= 3 && NSB >= 2 && NSB <= 4, "pipeline depths");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int SA = BM * 64, SM_ = BM * 8, ASTG = SA + SM_, SB = 64 * BN * 2;
  constexpr int BRING = 2 * ASTG;               // LDS: [A+metadata stage] x 2 | [B stage] x NSB
  constexpr int NBW = NLB > 0 ? NLB : WM * WN;   // waves that issue the B DMA: the B loaders, or the consumers themselves
  constexpr int B_N = BN / 8, B_WI = B_N / NBW;  // B DMA instructions per stage / per issuing wave
  static_assert(B_N % NBW == 0 && B_WI >= 1 && (NSB - 2) * B_WI <= 63, "B tile vs issuing waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
#ifdef SM_TUNING
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, (p.ablate & 8) != 0);  // (tuning: SM_DIRECT_ABLATE bit 3 = dispatch order; measured 0-5 % slower here, mma_tile.h)
#else
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
#endif
  const unsigned gb = lid / tiles, trem = lid - gb * tiles;
  const unsigned grp = gb / (unsigned)p.batch, b = gb - grp * (unsigned)p.batch;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = p.K / 64;
  half_t* C = p.C[grp] + (size_t)b * p.sC;

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned wm = wave / WN, wn = wave % WN;  // consumer waves only

  if (NLB > 0 && wave >= (unsigned)(NC + NLA)) {
    // ------------------------------------------------------------------ B loader wave: LDS-DMA only
    BTileDma<BN, NBW> bd;
    bd.setup(p.B[grp] + (size_t)b * p.sB, p.N, n0, wave - (NC + NLA), lane, BRING);
    auto issue = [&](int kt, int buf) {
#if defined(SM_ABLATE) && (SM_ABLATE & 4)
      return;  /* diagnostic timing builds only: 1 = no consumer compute, 2 = no selection, 4 = no B DMA, 8 = no A loads in the loop */
#endif
      bd.issue(smem, kt, (unsigned)(buf * SB));
    };
#pragma unroll
    for (int s = 0; s < NSB - 1; ++s)
      if (s < nkt) issue(s, s);
    int nb = NSB - 1;  // buffer of the next stage to issue
    SM_T(unsigned long long tb = 0, ti = 0, tv = 0, nlong = 0; unsigned long long s0 = sm_stamp(); unsigned long long sprev = s0;)
    for (int kt = 0; kt < nkt; ++kt) {
      // stage kt has landed once at most the NSB-2 younger stages are still in flight (fewer exist at the tail)
#ifdef SM_STAMP
      if (kt + NSB - 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSB - 2) * B_WI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long sv = sm_stamp();
      tv += sv - s0;
      asm volatile("s_barrier" ::: "memory");
#else
      if (kt + NSB - 2 < nkt) wait_dma_and_barrier<(NSB - 2) * B_WI>();
      else wait_dma_and_barrier<0>();
#endif
      SM_T(unsigned long long s1 = sm_stamp(); tb += s1 - s0;)
      if (kt + NSB - 1 < nkt) {
        issue(kt + NSB - 1, nb);  // the buffer stage kt-1 occupied: consumers left it before barrier kt
        nb = nb + 1 == NSB ? 0 : nb + 1;
      }
      SM_T(s0 = sm_stamp(); ti += s0 - s1; { const unsigned long long busy = (s0 - s1) + (sv - sprev); nlong += busy > 2000 ? 1 : 0; sprev = s0; })
    }
    SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8; d[0] = tb; d[1] = ti; d[2] = tv; d[3] = 2; d[6] = nlong; })
  } else if (wave >= (unsigned)NC) {
    // ------------------------------------------------------------------ A loader wave: load, select, ds_write
    // (round 4: raising these waves' priority -- their selection is the stage's critical path -- changed nothing: within 2 %
    //  either way on every wide shape, profiles/ab_variants_r04c.txt; three A stages in flight lost 3-6 %)
    const unsigned lw = wave - NC;
    const half_t* A = p.A[grp] + (size_t)b * p.sA;
    const int mlast = p.Mrows - 1;
    // load i of this wave = rows 8*(4*lw + i) + lane/8, dense chunk c = lane % 8 (k 8c .. 8c+7 of the stage)
    const unsigned c8 = lane & 7u;
    const half_t* a_src[4];
    unsigned a_val_off[4], a_meta_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned row = 8u * (4u * lw + i) + (lane >> 3);
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      a_src[i] = A + (size_t)gr * p.lda + 8u * c8;
      a_val_off[i] = row * 64u + 16u * ((c8 >> 1) ^ a64_swz(row)) + 8u * (c8 & 1u);
      a_meta_off[i] = SA + row * 8u + c8;
    }
    // PF stages of A stay in flight in registers (plain loads: the compiler's counted vmcnt guards each use).
    // Its wait insertion takes the strictest count over all paths that reach a use, so the steady-state loop
    // below is free of conditionals (every iteration issues its loads) and the last stages run in a peeled tail.
    u4 ra[PF][4];
    auto load_a = [&](int kt, u4 (&dst)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // non-temporal only when A is read once by the whole grid (one column tile): with several, the later tiles' reads of the same
        // rows should find them in L2 (round 5, as for the direct kernel: profiles/nt_ab_r05q.txt)
        if constexpr (ANT) dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(a_src[i] + (size_t)kt * 64));
        else dst[i] = *reinterpret_cast<const u4*>(a_src[i] + (size_t)kt * 64);
      }
      __builtin_amdgcn_sched_barrier(0);  // stages are issued in stage order on every path (the counted waits rely on it)
    };
    auto write_stage = [&](int kt, const u4 (&src)[4]) {
      char* sb = smem + (kt & 1) * ASTG;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t k0, k1, n0, n1;
#if defined(SM_ABLATE) && (SM_ABLATE & 2)
        k0 = src[i][0]; k1 = src[i][2]; n0 = 4; n1 = 4;
#else
        strip_select_f16(src[i][0], src[i][1], k0, n0);
        strip_select_f16(src[i][2], src[i][3], k1, n1);
#endif
        *reinterpret_cast<u2*>(sb + a_val_off[i]) = u2{k0, k1};
        *reinterpret_cast<unsigned char*>(sb + a_meta_off[i]) = (unsigned char)(n0 | (n1 << 4));
      }
    };
    SM_T(unsigned long long tb = 0, tw = 0, ti = 0, tv = 0, tmax = 0, nlong = 0; unsigned long long s0 = 0;)
    auto step = [&](int kt, u4 (&rr)[4], bool write, bool load) {
      // stage kt is complete in LDS once this wave's ds_writes have landed; loads stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);   // keep the next stage's selection (and its vmcnt wait) below the barrier
      SM_T(unsigned long long s1 = sm_stamp(); tb += s1 - s0;)
      if (write) {
        SM_T(asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (PF - 1)) : "memory"); unsigned long long s1b = sm_stamp(); tv += s1b - s1;)
        write_stage(kt + 1, rr);  // buffer (kt+1)&1: consumers left it before barrier kt
      }
      SM_T(__builtin_amdgcn_sched_barrier(0); unsigned long long s2 = sm_stamp(); tw += s2 - s1;)
#if !(defined(SM_ABLATE) && (SM_ABLATE & 8))
      if (load) load_a(kt + 1 + PF, rr);
#endif
      __builtin_amdgcn_sched_barrier(0);
      SM_T(s0 = sm_stamp(); ti += s0 - s2; { const unsigned long long busy = s0 - s1; tmax = busy > tmax ? busy : tmax; nlong += busy > 2000 ? 1 : 0; })
    };
    int kt0 = 0;
    SM_T(s0 = sm_stamp();)
    if (nkt > 2 * PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) load_a(s, ra[s]);
      write_stage(0, ra[0]);
      load_a(PF, ra[0]);
      for (; kt0 + 2 * PF < nkt; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) step(kt0 + u, ra[(u + 1) % PF], true, true);
      }
    } else {
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (s < nkt) load_a(s, ra[s]);
      write_stage(0, ra[0]);
      if (PF < nkt) load_a(PF, ra[0]);
    }
    for (; kt0 < nkt; kt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int kt = kt0 + u;
        if (kt < nkt) step(kt, ra[(u + 1) % PF], kt + 1 < nkt, kt + 1 + PF < nkt);
      }
    }
    SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8; d[0] = tb; d[1] = tw; d[2] = ti; d[3] = 1; d[4] = tv; d[5] = tmax; d[6] = nlong; })
  } else {
    // ------------------------------------------------------------------ consumer wave (as spmma_f16_pc_kernel)
    int cb = 0;
    // without B loader waves (n <= 128: eight waves per workgroup keep three workgroups on a CU) the consumers issue
    // the B tile's DMA themselves -- they have no other vector-memory traffic, so the counted vmcnt stays theirs --
    // and the loader waves only carry A, two stages ahead
    BTileDma<BN, NBW> bd;
    int nb = NSB - 1;
    if (NLB == 0) {
      bd.setup(p.B[grp] + (size_t)b * p.sB, p.N, n0, wave, lane, BRING);
#pragma unroll
      for (int s = 0; s < NSB - 1; ++s)
        if (s < nkt) bd.issue(smem, s, (unsigned)(s * SB));
    }
    SM_T(unsigned long long tb = 0, tc = 0, nlong = 0; unsigned long long s0 = sm_stamp();)
    for (int kt = 0; kt < nkt; ++kt) {
      if (NLB == 0) {
        if (kt + NSB - 2 < nkt) wait_dma_and_barrier<(NSB - 2) * B_WI>();
        else wait_dma_and_barrier<0>();
        if (kt + NSB - 1 < nkt) {
          bd.issue(smem, kt + NSB - 1, (unsigned)(nb * SB));
          nb = nb + 1 == NSB ? 0 : nb + 1;
        }
      } else {
        wait_dma_and_barrier<0>();
      }
      SM_T(unsigned long long s1 = sm_stamp(); tb += s1 - s0;)
#if defined(SM_ABLATE) && (SM_ABLATE & 1)
      cb = cb + 1 == NSB ? 0 : cb + 1;
      continue;
#endif
      const char* As = smem + (kt & 1) * ASTG;
      const char* Ms = As + SA;
      const char* Bs = smem + BRING + cb * SB;
      smfmac_stage<FM, FN, BF>(As, Ms, Bs, wm * TM, wn * TN, lane, acc);
      cb = cb + 1 == NSB ? 0 : cb + 1;
      SM_T(__builtin_amdgcn_sched_barrier(0); s0 = sm_stamp(); tc += s0 - s1; nlong += (s0 - s1) > 2000 ? 1 : 0;)
    }
    SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 8; d[0] = tb; d[1] = tc; d[2] = 0; d[3] = 0; d[6] = nlong; })
  }
  __syncthreads();

  store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, wave < (unsigned)NC, wm * TM, wn * TN, m0, n0, p.Mrows, p.N, p.alpha, p.beta, tid);
}

template <int BN, int WM, int WN, int NLB, int PF, int NSB, bool BF = false, bool ANT = true>
static int launch_fused_wide(const FusedArgs& a0, hipStream_t st) {
  FusedArgs a = a0;
#ifdef SM_TUNING
  if (!a.ablate) a.ablate = tuning_int("SM_DIRECT_ABLATE", 0) & 24;
#endif
  a.tiles_m = (a.Mrows + 127) / 128;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch * a.ngroup;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_spmma_fused_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = 2 * (size_t)128 * 72 + (size_t)NSB * 64 * BN * 2;
  constexpr size_t lds_epi = (size_t)128 * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static LdsOptIn lds_optin;
  if (lds > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_fused_wide_kernel<BN, WM, WN, NLB, PF, NSB, BF, ANT>), lds, "spmma_f16_fused_wide_kernel")) return rc;
  }
#ifdef SM_STAMP
  {
    constexpr int NWV = WM * WN + 4 + NLB;
    static unsigned long long* dbg = nullptr;
    static size_t cap = 0;
    const size_t cnt = nwg * (size_t)NWV * 8;
    if (cnt > cap) { if (dbg) (void)hipFree(dbg); (void)hipMalloc((void**)&dbg, cnt * 8); cap = cnt; }
    (void)hipMemset(dbg, 0, cnt * 8);
    a.dbg = dbg;
    spmma_f16_fused_wide_kernel<BN, WM, WN, NLB, PF, NSB, BF, ANT><<<dim3((unsigned)nwg), dim3(64 * NWV), lds, st>>>(a);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(cnt);
    (void)hipMemcpy(h.data(), dbg, cnt * 8, hipMemcpyDeviceToHost);
    double LBl = 0, Cnl = 0, LA[6] = {0, 0, 0, 0, 0, 0}, LB[3] = {0, 0, 0}, Cn[2] = {0, 0}, na = 0, nbw = 0, nc = 0;
    for (size_t i = 0; i < cnt / 8; ++i) {
      const unsigned long long* d = &h[i * 8];
      if (d[3] == 1) { LA[0] += d[0]; LA[1] += d[1]; LA[2] += d[2]; LA[3] += d[4]; LA[4] += d[5]; LA[5] += d[6]; na += 1; }
      else if (d[3] == 2) { LB[0] += d[0]; LB[1] += d[1]; LB[2] += d[2]; LBl += d[6]; nbw += 1; }
      else { Cn[0] += d[0]; Cn[1] += d[1]; Cnl += d[6]; nc += 1; }
    }
    const double nk = (double)(a.K / 64);
    fprintf(stderr, "STAMP-FUSED-WIDE %dx%dx%d PF=%d NSB=%d tiles=%zu nkt=%d | A-loader per stage: barrier %.0f vmcnt-wait %.0f select+write(incl. wait) %.0f load-issue %.0f | B-loader: wait+barrier %.0f (vmcnt part %.0f) issue %.0f | consumer: barrier %.0f compute %.0f\n",
            a.Mrows, a.N, a.K, PF, NSB, nwg, a.K / 64, LA[0] / na / nk, LA[3] / na / nk, LA[1] / na / nk, LA[2] / na / nk,
            LB[0] / nbw / nk, LB[2] / nbw / nk, LB[1] / nbw / nk, Cn[0] / nc / nk, Cn[1] / nc / nk);
    fprintf(stderr, "   A-loader busy per stage: mean of per-wave max %.0f, stages busier than 2000 cycles per wave %.1f of %d (B-loader %.1f, consumer %.1f)\n", LA[4] / na, LA[5] / na, a.K / 64, LBl / nbw, Cnl / nc);
    return check_launch("spmma_f16_fused_wide_kernel");
  }
#endif
  spmma_f16_fused_wide_kernel<BN, WM, WN, NLB, PF, NSB, BF, ANT><<<dim3((unsigned)nwg), dim3(64 * (WM * WN + 4 + NLB)), lds, st>>>(a);
  return check_launch("spmma_f16_fused_wide_kernel");
}

// ---------------------------------------------------------------------------------------------
// PERSISTENT form of the wide kernel (round 3).  One workgroup per CU walks its tiles (id = blockIdx.x + j * gridDim.x,
// XCD-remapped) with ONE stage counter that runs across tile boundaries: the A loaders' register pipeline, the A image's
// double buffer and the B ring never drain, so while the consumers write tile j's C the first stages of tile j + 1 are
// already in flight / in LDS (the plain kernel pays a workgroup teardown + launch + pipeline fill of ~3 stage times per
// tile: 8 % of a 36-stage tile, 25 % of an 8-stage one).  Roles, LDS images, instruction sequence per stage and therefore
// C are those of spmma_f16_fused_wide_kernel; what differs is the epilogue -- each consumer wave stores its 32 x 128 piece
// of C through a wave-private LDS patch in two 64-column halves (no workgroup barrier, nothing aliases the rings) -- and
// the per-tile pointer set-up, which moves into the loaders' issue cursors.  Every wave executes exactly one s_barrier
// per stage of the workgroup's S = tiles x K/64 stages.
// ---------------------------------------------------------------------------------------------
template <int BN, int WM, int WN, int NLB, int PF, int NSB, bool BF = false>
__global__ __launch_bounds__(64 * (WM * WN + 4 + NLB)) void spmma_f16_fused_widep_kernel(const FusedArgs p, const unsigned total) {
  constexpr int BM = 128, NLA = 4, NC = WM * WN;
  static_assert(NLB > 0 && PF >= 1 && PF <= 3 && NSB >= 2 && NSB <= 4, "pipeline depths");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(TN == 128 && TM == 32, "epilogue patch geometry");
  constexpr int SA = BM * 64, SM_ = BM * 8, ASTG = SA + SM_, SB = 64 * BN * 2;
  constexpr int BRING = 2 * ASTG;
  constexpr int B_N = BN / 8, B_WI = B_N / NLB;
  static_assert(B_N % NLB == 0 && (NSB - 2) * B_WI <= 63, "B tile vs loader waves");
  constexpr int WPITCH = 64 * 2 + 8, WPATCH = TM * WPITCH;  // wave-private C patch: 32 rows x (64 columns + pad)
  constexpr int CPATCH = BRING + NSB * SB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const int nkt = p.K / 64;
  const unsigned my_tiles = (total - blockIdx.x + gridDim.x - 1u) / gridDim.x;  // >= 1: gridDim.x <= total
  const int S = (int)my_tiles * nkt;

  struct Tile {
    unsigned grp, b;
    int m0, n0;
  };
  auto tile_of = [&](unsigned j) {
    Tile t;
    const unsigned lid = xcd_remap(blockIdx.x + j * gridDim.x, total);
    const unsigned gb = lid / tiles, trem = lid - gb * tiles;
    t.grp = gb / (unsigned)p.batch;
    t.b = gb - t.grp * (unsigned)p.batch;
    const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
    t.m0 = (int)tile_m * BM;
    t.n0 = (int)tile_n * BN;
    return t;
  };

  if (wave >= (unsigned)(NC + NLA)) {
    // ------------------------------------------------------------------ B loader wave: LDS-DMA only
    BTileDma<BN, NLB> bd;
    unsigned ij = 0;  // issue cursor: tile and stage of the next B stage to issue
    int ikt = 0, nb = 0;
    {
      const Tile t = tile_of(0);
      bd.setup(p.B[t.grp] + (size_t)t.b * p.sB, p.N, t.n0, wave - (NC + NLA), lane, BRING);
    }
    auto issue = [&]() {
      bd.issue(smem, ikt, (unsigned)(nb * SB));
      nb = nb + 1 == NSB ? 0 : nb + 1;
      if (++ikt == nkt) {
        ikt = 0;
        if (++ij < my_tiles) {
          const Tile t = tile_of(ij);
          bd.setup(p.B[t.grp] + (size_t)t.b * p.sB, p.N, t.n0, wave - (NC + NLA), lane, BRING);
        }
      }
    };
#pragma unroll
    for (int s = 0; s < NSB - 1; ++s)
      if (s < S) issue();
    for (int g = 0; g < S; ++g) {
      if (g + NSB - 2 < S) wait_dma_and_barrier<(NSB - 2) * B_WI>();
      else wait_dma_and_barrier<0>();
      if (g + NSB - 1 < S) issue();  // into the buffer stage g - 1 occupied: the consumers left it before barrier g
    }
  } else if (wave >= (unsigned)NC) {
    // ------------------------------------------------------------------ A loader wave: load, select, ds_write
    const unsigned lw = wave - NC;
    const unsigned c8 = lane & 7u;
    const half_t* a_src[4];
    unsigned a_val_off[4], a_meta_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned row = 8u * (4u * lw + i) + (lane >> 3);
      a_val_off[i] = row * 64u + 16u * ((c8 >> 1) ^ a64_swz(row)) + 8u * (c8 & 1u);
      a_meta_off[i] = SA + row * 8u + c8;
    }
    const int mlast = p.Mrows - 1;
    auto point_at = [&](unsigned j) {  // row pointers of this wave's four loads in tile j
      const Tile t = tile_of(j);
      const half_t* A = p.A[t.grp] + (size_t)t.b * p.sA;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned row = 8u * (4u * lw + i) + (lane >> 3);
        int gr = t.m0 + (int)row;
        gr = gr < mlast ? gr : mlast;
        a_src[i] = A + (size_t)gr * p.lda + 8u * c8;
      }
    };
    point_at(0);
    unsigned lj = 0;  // load cursor (stages are loaded strictly in order)
    int lkt = 0;
    u4 ra[PF][4];
    auto load_a = [&](u4 (&dst)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(a_src[i] + (size_t)lkt * 64));
      __builtin_amdgcn_sched_barrier(0);
      if (++lkt == nkt) {
        lkt = 0;
        if (++lj < my_tiles) point_at(lj);
      }
    };
    auto write_stage = [&](int g, const u4 (&src)[4]) {
      char* sb = smem + (g & 1) * ASTG;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t k0, k1, n0, n1;
        strip_select_f16(src[i][0], src[i][1], k0, n0);
        strip_select_f16(src[i][2], src[i][3], k1, n1);
        *reinterpret_cast<u2*>(sb + a_val_off[i]) = u2{k0, k1};
        *reinterpret_cast<unsigned char*>(sb + a_meta_off[i]) = (unsigned char)(n0 | (n1 << 4));
      }
    };
    auto step = [&](int g, u4 (&rr)[4], bool write, bool load) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (write) write_stage(g + 1, rr);  // buffer (g+1)&1: the consumers left it before barrier g
      if (load) load_a(rr);               // stage g + 1 + PF
      __builtin_amdgcn_sched_barrier(0);
    };
    int g0 = 0;
    if (S > 2 * PF) {
#pragma unroll
      for (int s = 0; s < PF; ++s) load_a(ra[s]);
      write_stage(0, ra[0]);
      load_a(ra[0]);
      for (; g0 + 2 * PF < S; g0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) step(g0 + u, ra[(u + 1) % PF], true, true);
      }
    } else {
#pragma unroll
      for (int s = 0; s < PF; ++s)
        if (s < S) load_a(ra[s]);
      write_stage(0, ra[0]);
      if (PF < S) load_a(ra[0]);
    }
    for (; g0 < S; g0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int g = g0 + u;
        if (g < S) step(g, ra[(u + 1) % PF], g + 1 < S, g + 1 + PF < S);
      }
    }
  } else {
    // ------------------------------------------------------------------ consumer wave
    const unsigned g4 = lane >> 4, r = lane & 15u;
    const unsigned wm = wave / WN, wn = wave % WN;
    char* const patch = smem + CPATCH + wave * WPATCH;
    f4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    int cb = 0, kt = 0;
    unsigned cj = 0;
    Tile t = tile_of(0);
    for (int g = 0; g < S; ++g) {
      asm volatile("s_barrier" ::: "memory");  // (no vmcnt: this wave's C stores of the previous tile may still be in flight)
      const char* As = smem + (g & 1) * ASTG;
      smfmac_stage<FM, FN, BF>(As, As + SA, smem + BRING + cb * SB, wm * TM, wn * TN, lane, acc);
      cb = cb + 1 == NSB ? 0 : cb + 1;
      if (++kt == nkt) {
        // ---- this tile is done: the wave's 32 x 128 piece of C, two 64-column halves through its private patch
        half_t* C = p.C[t.grp] + (size_t)t.b * p.sC;
        const bool c_vec = (reinterpret_cast<uintptr_t>(C) & 15u) == 0 && (p.N % 8 == 0);
        if (p.beta == 0.0f && c_vec) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
              for (int j = 0; j < FN / 2; ++j) {
                const unsigned lr = i * 16 + 4u * g4, lc = j * 16 + r;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  *reinterpret_cast<half_t*>(patch + (lr + q) * WPITCH + lc * 2) = to_elt<BF>(p.alpha * acc[i][h * (FN / 2) + j][q]);
              }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave wrote and reads: no barrier
            const int gc = t.n0 + (int)(wn * TN) + 64 * h + 8 * (int)(lane & 7u);
#pragma unroll
            for (int s8 = 0; s8 < TM / 8; ++s8) {
              const unsigned lr = 8u * s8 + (lane >> 3);
              const int gr = t.m0 + (int)(wm * TM + lr);
              if (gr < p.Mrows && gc < p.N) {
                const u2 lo = *reinterpret_cast<const u2*>(patch + lr * WPITCH + 16u * (lane & 7u));
                const u2 hi = *reinterpret_cast<const u2*>(patch + lr * WPITCH + 16u * (lane & 7u) + 8u);
                __builtin_nontemporal_store(u4{lo[0], lo[1], hi[0], hi[1]}, reinterpret_cast<u4*>(C + (size_t)gr * p.N + gc));
              }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the patch is rewritten by the next half / tile
          }
        } else {  // beta != 0 or a C that cannot take 16-byte stores: element-wise, reading C once
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              const int gc = t.n0 + (int)(wn * TN + j * 16 + r);
              if (gc >= p.N) continue;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int gr = t.m0 + (int)(wm * TM + i * 16 + 4u * g4) + q;
                if (gr >= p.Mrows) continue;
                half_t* dst = C + (size_t)gr * p.N + gc;
                float v = p.alpha * acc[i][j][q];
                if (p.beta != 0.0f) v += p.beta * to_f32<BF>(*dst);
                *dst = to_elt<BF>(v);
              }
            }
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        kt = 0;
        if (++cj < my_tiles) t = tile_of(cj);
      }
    }
  }
}

template <int BN, int WM, int WN, int NLB, int PF, int NSB, bool BF = false>
static int launch_fused_widep(const FusedArgs& a0, hipStream_t st) {
  FusedArgs a = a0;
#ifdef SM_TUNING
  if (!a.ablate) a.ablate = tuning_int("SM_DIRECT_ABLATE", 0) & 24;
#endif
  a.tiles_m = (a.Mrows + 127) / 128;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nt = (size_t)a.tiles_m * a.tiles_n * a.batch * a.ngroup;
  if (nt == 0) return SM_STATUS_SUCCESS;
  if (nt > 0x7fffffffu) {
    set_error("sm_spmma_fused_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds = 2 * (size_t)128 * 72 + (size_t)NSB * 64 * BN * 2 + (size_t)(WM * WN) * 32 * (64 * 2 + 8);
  static_assert(lds <= 160 * 1024, "LDS budget of the persistent wide kernel");
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_fused_widep_kernel<BN, WM, WN, NLB, PF, NSB, BF>), lds, "spmma_f16_fused_widep_kernel")) return rc;
  // one workgroup per CU (LDS); ids congruent mod 8 stay on one XCD across a workgroup's tiles when the grid is a multiple of 8
  size_t cus = (size_t)device_cu_count();
  cus -= cus % 8;
  if (cus == 0) cus = 8;
  const unsigned grid = (unsigned)(nt <= cus ? nt : cus);
  spmma_f16_fused_widep_kernel<BN, WM, WN, NLB, PF, NSB, BF><<<dim3(grid), dim3(64 * (WM * WN + 4 + NLB)), lds, st>>>(a, (unsigned)nt);
  return check_launch("spmma_f16_fused_widep_kernel");
}

// ---------------------------------------------------------------------------------------------
// n > 256 with a short K (k <= 512): A-stationary.  The 2:4 image of the workgroup's 128 rows for the WHOLE K
// (k/64 stages x 9 KiB) stays in LDS; all 16 waves build it first (every load of the panel in flight at once: a single
// HBM latency), 4 of them then only attend the barriers, and the workgroup walks its column tiles re-using it: A is loaded and selected once per `nsplit` column range
// instead of once per column tile, and from the second tile on a stage costs only the B tile's 128 line requests.
// B streams through its own ring across tile boundaries (no drain); each consumer wave stores its own 32 x 64
// piece of C through a wave-private LDS patch (no barrier in the epilogue), so the loaders keep prefetching the
// next tile's B while C is written.  beta == 0 only (the caller falls back to the wide kernel otherwise).
// ---------------------------------------------------------------------------------------------
template <int NSB, bool BF = false>
__global__ __launch_bounds__(64 * 16) void spmma_f16_fused_astat_kernel(const FusedArgs p, int nsplit, int tiles_per_split) {
  constexpr int BM = 128, BN = 128, WM = 4, WN = 2, NC = 8, NLA = 4, NLB = 4;
  static_assert(NC + NLA + NLB == 16, "launch bounds");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;  // wave tile 32 x 64
  constexpr int SA = BM * 64, SM_ = BM * 8, ASTG = SA + SM_, SB = 64 * BN * 2;
  constexpr int B_N = BN / 8, B_WI = B_N / NLB;
  static_assert((NSB - 2) * B_WI <= 63, "vmcnt range");
  // wave-private C patch: 32 rows x 136 B.  34 dwords per row: the four 4-row groups a ds_write_b16 instruction
  // touches (rows 4g + q) start 8 banks apart, each covering 8 banks: no conflict beyond the two halves of a dword
  constexpr int WPITCH = TN * 2 + 8, WPATCH = TM * WPITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkt = p.K / 64;
  const unsigned per_batch = (unsigned)p.tiles_m * (unsigned)nsplit;
#ifdef SM_TUNING
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, (p.ablate & 8) != 0);  // (tuning: SM_DIRECT_ABLATE bit 3 = dispatch order; measured 0-5 % slower here, mma_tile.h)
#else
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
#endif
  const unsigned gb = lid / per_batch, trem = lid - gb * per_batch;
  const unsigned grp = gb / (unsigned)p.batch, b = gb - grp * (unsigned)p.batch;
  const unsigned tile_m = trem / (unsigned)nsplit, split = trem - tile_m * (unsigned)nsplit;
  const int m0 = (int)tile_m * BM;
  const int nt0 = (int)split * tiles_per_split;
  const int nt1 = nt0 + tiles_per_split < p.tiles_n ? nt0 + tiles_per_split : p.tiles_n;
  const int T = (nt1 - nt0) * nkt;  // stage iterations of this workgroup (>= nkt: the launcher never makes empty splits)
  char* const Bring = smem + nkt * ASTG;
  char* const Cpatch = Bring + NSB * SB;
  half_t* C = p.C[grp] + (size_t)b * p.sC;

  // B loader state (waves NC+NLA ..): set up first so that their ring prologue is in flight during phase 1
  const bool is_b = wave >= (unsigned)(NC + NLA);
  const unsigned lwb = is_b ? wave - (NC + NLA) : 0u;
  const char* Bb = reinterpret_cast<const char*>(p.B[grp] + (size_t)b * p.sB);
  unsigned b_kr[B_WI], b_col[B_WI], b_dst[B_WI];
#pragma unroll
  for (int i = 0; i < B_WI; ++i) {
    const unsigned j = lwb + (unsigned)NLB * i, panel = j >> 3, grp = j & 7u, kr = 8u * grp + (lane >> 3);
    b_kr[i] = kr;
    b_col[i] = 64u * panel + 8u * ((lane & 7u) ^ b_swz(kr));
    b_dst[i] = panel * 8192u + grp * 1024u;
  }
  const size_t row_bytes = (size_t)p.N * 2;
  int i_nt = nt0, i_kt = 0, i_buf = 0;  // the next B stage to issue
  auto issue_b = [&]() {
#if defined(SM_ABLATE) && (SM_ABLATE & 4)
    if (nkt > 0) { i_buf = i_buf + 1 == NSB ? 0 : i_buf + 1; if (++i_kt == nkt) { i_kt = 0; ++i_nt; } return; }
#endif
#pragma unroll
    for (int i = 0; i < B_WI; ++i) {
      int gc = i_nt * BN + (int)b_col[i];
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      const char* src = Bb + (size_t)(i_kt * 64 + (int)b_kr[i]) * row_bytes + (size_t)gc * 2;
      __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(Bring + i_buf * SB + b_dst[i]), 16, 0, 0);
    }
    i_buf = i_buf + 1 == NSB ? 0 : i_buf + 1;
    if (++i_kt == nkt) { i_kt = 0; ++i_nt; }
  };
  if (is_b) {
#pragma unroll
    for (int s = 0; s < NSB - 1; ++s)
      if (s < T) issue_b();
  }

  // ---- phase 1, all 16 waves: the 2:4 image of the row panel for the whole K.  Wave w owns rows 8w .. 8w+7: one
  // 16-byte load per lane and stage, all stages in flight at once (K <= 512: at most 8 loads), then selection.
  {
    const half_t* A = p.A[grp] + (size_t)b * p.sA;
    const unsigned row = 8u * wave + (lane >> 3), c8 = lane & 7u;
    int gr = m0 + (int)row;
    gr = gr < p.Mrows - 1 ? gr : p.Mrows - 1;
    const half_t* a_src = A + (size_t)gr * p.lda + 8u * c8;
    const unsigned a_val_off = row * 64u + 16u * ((c8 >> 1) ^ a64_swz(row)) + 8u * (c8 & 1u);
    const unsigned a_meta_off = SA + row * 8u + c8;
    u4 ra[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < nkt) ra[s] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(a_src + (size_t)s * 64));
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (s < nkt) {
        uint32_t k0, k1, n0, n1;
        strip_select_f16(ra[s][0], ra[s][1], k0, n0);
        strip_select_f16(ra[s][2], ra[s][3], k1, n1);
        char* sb = smem + s * ASTG;
        *reinterpret_cast<u2*>(sb + a_val_off) = u2{k0, k1};
        *reinterpret_cast<unsigned char*>(sb + a_meta_off) = (unsigned char)(n0 | (n1 << 4));
      }
  }
  __syncthreads();  // the image is complete (and the B prologue has landed)
  if (wave >= (unsigned)NC && !is_b) {  // 4 of the 16 waves were only needed for phase 1: they just keep the barrier count
    // (letting them write C out of the consumers' patches was tried: no gain, the consumers' 32 ds_write_b16 per tile
    // and the stage barriers, not the store queue, set the tile's time)
    for (int it = 0; it < T; ++it) asm volatile("s_barrier" ::: "memory");
    return;
  }

  if (is_b) {
    // ------------------------------------------------------------------ B loader wave
    SM_T(unsigned long long tv = 0, tb = 0, ti = 0; unsigned long long s0 = sm_stamp();)
    for (int it = 0; it < T; ++it) {
#ifdef SM_STAMP
      if (it + NSB - 2 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSB - 2) * B_WI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long sv = sm_stamp();
      tv += sv - s0;
      asm volatile("s_barrier" ::: "memory");
      const unsigned long long s1 = sm_stamp();
      tb += s1 - sv;
#else
      if (it + NSB - 2 < T) wait_dma_and_barrier<(NSB - 2) * B_WI>();
      else wait_dma_and_barrier<0>();
#endif
      if (it + NSB - 1 < T) issue_b();
      SM_T(s0 = sm_stamp(); ti += s0 - s1;)
    }
    SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 16 + wave) * 8; d[0] = tv; d[1] = tb; d[2] = ti; d[3] = 2; d[4] = (unsigned long long)T; })
  } else {
    // ------------------------------------------------------------------ consumer wave
    const unsigned g = lane >> 4, r = lane & 15u;
    const unsigned wm = wave / WN, wn = wave % WN;
    char* const patch = Cpatch + wave * WPATCH;
    f4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    int cb = 0, kt = 0, nt = nt0;
    SM_T(unsigned long long tb = 0, tc = 0, te = 0; unsigned long long s0 = sm_stamp();)
    for (int it = 0; it < T; ++it) {
      asm volatile("s_barrier" ::: "memory");  // (no vmcnt: this wave's C stores of the previous tile may still be in flight)
      SM_T(unsigned long long s1 = sm_stamp(); tb += s1 - s0;)
#if defined(SM_ABLATE) && (SM_ABLATE & 1)
      if (nkt > 0) { cb = cb + 1 == NSB ? 0 : cb + 1; if (++kt == nkt) { kt = 0; ++nt; } continue; }
#endif
      const char* As = smem + kt * ASTG;
      const char* Ms = As + SA;
      const char* Bs = Bring + cb * SB;
      smfmac_stage<FM, FN, BF>(As, Ms, Bs, wm * TM, wn * TN, lane, acc);
      cb = cb + 1 == NSB ? 0 : cb + 1;
      SM_T(__builtin_amdgcn_sched_barrier(0); s0 = sm_stamp(); tc += s0 - s1;)
#if defined(SM_ABLATE) && (SM_ABLATE & 16)
      if (++kt == nkt) { kt = 0; ++nt; }
      if (nkt > 0) continue;
#endif
      if (++kt == nkt) {
        // ---- this column tile is done: C piece of this wave through its private LDS patch, 16-byte row pieces out
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const unsigned lr = i * 16 + 4u * g, lc = j * 16 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              *reinterpret_cast<half_t*>(patch + (lr + q) * WPITCH + lc * 2) = to_elt<BF>(p.alpha * acc[i][j][q]);
              acc[i][j][q] = 0.f;
            }
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave wrote and reads: no barrier
        const int gc = nt * BN + (int)(wn * TN) + 8 * (int)(lane & 7u);
#pragma unroll
        for (int s = 0; s < TM / 8; ++s) {
          const unsigned lr = 8u * s + (lane >> 3);
          const int gr = m0 + (int)(wm * TM + lr);
          if (gr < p.Mrows && gc < p.N) {
            const u2 lo = *reinterpret_cast<const u2*>(patch + lr * WPITCH + 16u * (lane & 7u));  // rows are 8-byte aligned
            const u2 hi = *reinterpret_cast<const u2*>(patch + lr * WPITCH + 16u * (lane & 7u) + 8u);
            __builtin_nontemporal_store(u4{lo[0], lo[1], hi[0], hi[1]}, reinterpret_cast<u4*>(C + (size_t)gr * p.N + gc));
          }
        }
        kt = 0;
        ++nt;
        SM_T(__builtin_amdgcn_sched_barrier(0); { const unsigned long long s2 = sm_stamp(); te += s2 - s0; s0 = s2; })
      }
    }
    SM_T(if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 16 + wave) * 8; d[0] = tb; d[1] = tc; d[2] = te; d[3] = 0; d[4] = (unsigned long long)T; })
  }
}

// (A PERSISTENT form of this kernel -- one workgroup per CU walking its row panels with two panel images in LDS, the four
// barrier-only waves building panel j + 1's image during panel j's stages, B ring across panels -- was built in round 3:
// bit-identical, and 10-18 % SLOWER than one workgroup per panel with either a 1- or a 4-interval prefetch ring
// (profiles/astatp_r03m.txt: 3136 x 512 x 128 x 4: 158.6 vs 140.2 us; 784 x 1024 x 256 x 6: 147.6 vs 124.9 us).  As with the
// wide kernel's long-K tiles, statically assigned work loses to the hardware's dynamic dispatch.  Removed again)

static size_t astat_lds_bytes(int nkt, int nsb) { return (size_t)nkt * (128 * 72) + (size_t)nsb * 64 * 128 * 2 + 8 * 32 * (64 * 2 + 8); }

template <bool BF = false>
static int launch_fused_astat(const FusedArgs& a0, hipStream_t st) {
  FusedArgs a = a0;
#ifdef SM_TUNING
  if (!a.ablate) a.ablate = tuning_int("SM_DIRECT_ABLATE", 0) & 24;
#endif
  a.tiles_m = (a.Mrows + 127) / 128;
  a.tiles_n = (a.N + 127) / 128;
  // split the column range of a row panel over several workgroups until 3/4 of the CUs have one (a split re-selects
  // its A panel, so no more than needed: 784 x 1024 x 256, b = 32: 29 / 33 / 41 us with 1 / 2 / 4 splits); >= 2 tiles
  // per split
  const int cus = device_cu_count();
  const size_t panels = (size_t)a.tiles_m * a.batch * a.ngroup;
  static const int nsplit_env = tuning_int("SM_FUSED_NSPLIT", 0);  // tuning aid
  int nsplit = 1;
  while (!nsplit_env && panels * nsplit * 4 < (size_t)3 * cus && (a.tiles_n + 2 * nsplit - 1) / (2 * nsplit) >= 2) nsplit *= 2;
  if (nsplit_env > 0) nsplit = nsplit_env < a.tiles_n ? nsplit_env : a.tiles_n;
  int tps = (a.tiles_n + nsplit - 1) / nsplit;
  nsplit = (a.tiles_n + tps - 1) / tps;  // no empty splits
  const size_t nwg = panels * nsplit;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_spmma_fused_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr int NSB = 3;  // (a ring of 4 measured the same)
  const size_t lds = astat_lds_bytes(a.K / 64, NSB);
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_fused_astat_kernel<NSB, BF>), (160 * 1024), "spmma_f16_fused_astat_kernel")) return rc;
  spmma_f16_fused_astat_kernel<NSB, BF><<<dim3((unsigned)nwg), dim3(64 * 16), lds, st>>>(a, nsplit, tps);
  return check_launch("spmma_f16_fused_astat_kernel");
}

}  // namespace sm

using namespace sm;

// Which shapes the two exact forms take -- ONE statement of each rule, used by spmma_fused16 below and by sm::spmma_fused16_takes_exact
// (the prune-in-place + multiply entry points ask BEFORE they modify A, csrc/spmma_f16_pruned.hip).
// span form: rows that are not whole 64-deep stages of 16-byte pieces, when A is one tall contiguous matrix, n <= 128 and a 128-row
// span + the whole B fit the LDS (k = 147: the stem layer of every ResNet)
static bool span_form_takes(size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC, bool all_aligned, bool c_aligned) {
  return (k % 64 != 0 || lda % 8 != 0) && lda == k && n % 8 == 0 && n <= 128 && all_aligned && c_aligned && (batch == 1 || (strideB == 0 && strideA == m * lda && strideC == m * n)) &&
         (m * batch * k * 2) % 16 == 0 && m * batch <= 0x7fffffffull && k <= 0x7fffffffull &&
         ((size_t)128 * k * 2 + 1152) + ((k + 63) / 64 * 64) * (n <= 64 ? 64 : 128) * 2 <= 160 * 1024;
}
// whole 64-deep stages of 16-byte aligned rows (every other kernel of this file)
static bool stage_forms_take(size_t n, size_t k, size_t lda, size_t strideA, size_t strideB, bool all_aligned) {
  return !(k % 64 != 0 || lda % 8 != 0 || strideA % 8 != 0 || n % 8 != 0 || strideB % 8 != 0 || !all_aligned);
}
static bool dims_fit_int(size_t m, size_t n, size_t k, size_t lda, size_t batch) {
  return !(m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || lda > 0x7fffffffull);
}
bool sm::spmma_fused16_takes_exact(const void* A, const void* B, const void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB,
                                   size_t strideC) {
  if (!A || !B || !C || lda < k || m == 0 || n == 0 || batch == 0) return false;
  if (n < 8 && k <= 64) return false;  // the thin form: inside the tight bound of the product, not the staged pair's bits
  const bool all_aligned = aligned16(A) && aligned16(B), c_aligned = aligned16(C);
  if (span_form_takes(m, n, k, lda, batch, strideA, strideB, strideC, all_aligned, c_aligned)) return true;
  return stage_forms_take(n, k, lda, strideA, strideB, all_aligned) && dims_fit_int(m, n, k, lda, batch);
}

template <bool BF>
static int spmma_fused16(size_t ngroup, const void* const* Ag, const void* const* Bg, void* const* Cg, size_t m, size_t n, size_t k, size_t lda,
                         size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                         sm_stream_t stream, void* workspace = nullptr, size_t workspace_bytes = 0) {
  if (ngroup == 0) return SM_STATUS_SUCCESS;
  if (!Ag || !Bg || !Cg || lda < k || ngroup > (size_t)MAXG) {
    set_error("sm_spmma_fused_{f16,bf16}: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  bool all_aligned = true, c_aligned = true;
  for (size_t g = 0; g < ngroup; ++g) {
    if (!Ag[g] || !Bg[g] || !Cg[g]) {
      set_error("sm_spmma_fused_{f16,bf16}: invalid argument (null operand)");
      return SM_STATUS_INVALID_VALUE;
    }
    all_aligned = all_aligned && aligned16(Ag[g]) && aligned16(Bg[g]);
    c_aligned = c_aligned && aligned16(Cg[g]);
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  // (round 5) thin problems -- n < 8, k <= 64, one tall contiguous A, shared B: the depthwise layers of the model zoo -- on the
  // vector ALUs (spmma_f16_thin.hip); result inside the tight bound of the exact product, not bit-identical to the staged pair
  if (n < 8 && k <= 64 && lda == k && (batch == 1 || (strideB == 0 && strideA == m * lda && strideC == m * n))) {
    const int rc = spmma_fused_thin(BF, (int)ngroup, Ag, Bg, Cg, m * batch, n, k, alpha, beta, (hipStream_t)stream);
    if (rc != SM_STATUS_NOT_SUPPORTED) return rc;
  }
  // rows that are not whole 64-deep stages of 16-byte pieces: the span form, when A is one tall contiguous matrix, n <= 128
  // and a 128-row span + the whole B fit the LDS (k = 147: the stem layer of every ResNet)
  if (span_form_takes(m, n, k, lda, batch, strideA, strideB, strideC, all_aligned, c_aligned)) {
    FusedArgs a = {};
    for (size_t g = 0; g < (size_t)MAXG; ++g) {
      const size_t s_ = g < ngroup ? g : 0;
      a.A[g] = (const half_t*)Ag[s_]; a.B[g] = (const half_t*)Bg[s_]; a.C[g] = (half_t*)Cg[s_];
    }
    a.ngroup = (int)ngroup;
    a.Mrows = (int)(m * batch); a.N = (int)n; a.K = (int)k; a.lda = (int)lda;
    a.batch = 1; a.alpha = alpha; a.beta = beta;
    hipStream_t st = (hipStream_t)stream;
    return n <= 64 ? launch_fused_span<64, BF>(a, st) : launch_fused_span<128, BF>(a, st);
  }
  // whole 64-deep stages of 16-byte aligned rows only; anything else: sm_compress24_f16 + sm_spmma_f16
  if (!stage_forms_take(n, k, lda, strideA, strideB, all_aligned)) {
    set_error("sm_spmma_fused_{f16,bf16}: needs k %% 64 == 0, n %% 8 == 0 and 16-byte aligned rows (use the staged path)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  if (ngroup > 1 && !c_aligned) {  // (the kernels' scalar C path keys on the pointer: one decision per launch)
    set_error("sm_spmma_fused_{f16,bf16}_grouped: needs 16-byte aligned C operands");
    return SM_STATUS_NOT_SUPPORTED;
  }
  if (!dims_fit_int(m, n, k, lda, batch)) {
    set_error("sm_spmma_fused_{f16,bf16}: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  FusedArgs a = {};
  for (size_t g = 0; g < (size_t)MAXG; ++g) {  // unused slots repeat problem 0 (never indexed: grp < ngroup)
    const size_t s_ = g < ngroup ? g : 0;
    a.A[g] = (const half_t*)Ag[s_]; a.B[g] = (const half_t*)Bg[s_]; a.C[g] = (half_t*)Cg[s_];
  }
  a.ngroup = (int)ngroup;
  void* const C = Cg[0];
  a.sA = strideA; a.sB = strideB; a.sC = strideC;
  a.Mrows = (int)m; a.N = (int)n; a.K = (int)k; a.lda = (int)lda;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  if (batch > 1 && strideB == 0 && strideA == m * lda && strideC == m * n) {
    a.Mrows = (int)(m * batch);
    a.batch = 1;
  }
  hipStream_t st = (hipStream_t)stream;
  // n <= 128 (and n <= 256 with a single stage): the direct kernel -- dense A by LDS-DMA, selection in the consumer's
  // registers, ring of 2 so that three workgroups share a CU.  SM_FUSED_DIRECT=3 (tuning aid): ring of 3.
  static const int direct_env = tuning_int("SM_FUSED_DIRECT", 2);
  static const int wide_env = tuning_int("SM_FUSED_WIDE", 0);  // tuning aid: force the wide kernel
  // (round 4) n > 128: the 256-row BIG form where its one-per-CU workgroups fill the chip's rounds at least as well as the
  // 128-row kernels' do (profiles/ab_big_r04a.txt: it then wins 3-10 %: 784 x 256 x 1024 x 5, 3136 x 256 x 512, 196 x 512 x 4608 x 3,
  // 784 x 512 x 1024; it loses where halving the tile count empties the last round: 784 x 256 x 2304 x 6 = 588 tiles = 2.3
  // rounds against 4.6, 196 x 512 x 2048 x 2 = 100 tiles against 196).  Against the A-stationary kernel (n > 256, k <= 512) it
  // needs a clear margin (196 x 2048 x 512 x 3: 59 vs 68 us; 784 x 1024 x 256 x 6: 126 vs 111).  SM_FUSED_BIG (tuning): 0 = never.
  const int big_rule = tuning_int("SM_FUSED_BIG", 8);
  auto round_eff = [](size_t tiles, size_t cus) { const size_t r = (tiles + cus - 1) / cus; return r ? (double)tiles / (double)(r * cus) : 1.0; };
  // (round 5) with a workspace: the STREAM-K form where the rounds of whole tiles leave CUs idle.  Costs in stages of a 256 x 256
  // tile: stream-K = its share of the stage units + ~3 stages for the fix-up of the two tiles a range cuts; the big form = rounds x
  // stages; a 128 x 256 tile's stage (wide kernel) takes ~0.6 of a 256 x 256 one (profiles/stamp_r04b_big_wide_direct.txt).  The
  // A-stationary shapes (n > 256, k <= 512) keep their kernel: few stages per tile, nothing for a K split to balance.
  // SM_FUSED_SK (tuning): 0 = never, 2 = wherever the kernel takes the shape.
  if (workspace && !wide_env) {
    const bool astat_shape = n > 256 && k <= 512 && beta == 0.0f && aligned16(C) && (strideC % 8 == 0);
    SkPlan pl;
    if (sk_takes((size_t)a.Mrows, (size_t)a.batch * a.ngroup, n, k, astat_shape, pl) &&
        workspace_bytes >= sk_workspace_bytes((size_t)pl.slots * ((n + 255) / 256), 256) && aligned16(workspace)) {
      //  the stream-K launcher's own limits (32-bit tile offsets, panel and workgroup counts) are "not taken", not a failure
      // of the call: a _ws call must succeed wherever the same call without a workspace does -- fall through to the forms below
      const int rc = n <= 256 ? launch_fused_sk<256, BF, true>(a, workspace, workspace_bytes, st) : launch_fused_sk<256, BF, false>(a, workspace, workspace_bytes, st);
      if (rc != SM_STATUS_NOT_SUPPORTED) return rc;
    }
  }
#ifdef SM_TUNING
  // (round 6, the residency experiment VERDICT round 5 asked for -- tuning builds only: SM_FUSED_D256=1) the few-tile shapes on 128 x 256 DIRECT
  // tiles: four waves, a ring of two 48 KiB stages = 96 KiB, so that a 48-64 KiB direct workgroup of ANOTHER launch fits on the same CU beside
  // it (the big / wide / A-stationary workgroups hold the whole LDS or 16 waves).  Measured with tools/overlap_probe.py: DESIGN.md 4.2.
  if (tuning_int("SM_FUSED_D256", 0) && n > 128 && k > 64) return launch_fused_direct<256, 2, BF, 128, 4, false>(a, st);
#endif
  if (big_rule == 8 && !wide_env && n > 128 && k > 64) {
    const size_t cus = (size_t)device_cu_count(), nb = (size_t)a.batch * a.ngroup;
    const size_t t_big = ((size_t)a.Mrows + 255) / 256 * ((n + 255) / 256) * nb, t_wide = ((size_t)a.Mrows + 127) / 128 * ((n + 255) / 256) * nb;
    const bool astat_shape = n > 256 && k <= 512 && beta == 0.0f && aligned16(C) && (strideC % 8 == 0);
    bool big = round_eff(t_big, cus) >= round_eff(t_wide, cus);
    if (astat_shape) {
      const size_t panels = ((size_t)a.Mrows + 127) / 128 * nb;
      size_t ns = 1, tn = (n + 127) / 128;
      while (panels * ns * 4 < 3 * cus && (tn + 2 * ns - 1) / (2 * ns) >= 2) ns *= 2;
      big = round_eff(t_big, cus) > round_eff(panels * ns, cus) + 0.1;
    }
    // (the hint on the big form's A loads, n <= 256: measured indifferent, profiles/nt_ab_r05q.txt)
    if (big) return n <= 256 ? launch_fused_big<256, BF, 3, 2, true>(a, st) : launch_fused_big<256, BF, 3, 2, false>(a, st);
  }
#ifdef SM_TUNING
  {  // A/B of the 256-row big form: bit 0 = n >= 256 (k > 64), bit 1 = 64 < n <= 128, bit 2 = n <= 256 with k <= 64; SM_FUSED_BIG_NSB = 2 / 3
    const int big_env = tuning_int("SM_FUSED_BIG", 0), nsb = tuning_int("SM_FUSED_BIG_NSB", 2);
    if (big_env < 8 && (big_env & 1) && n > 128 && k > 64) return n <= 256 ? launch_fused_big<256, BF, 3, 2, true>(a, st) : launch_fused_big<256, BF, 3, 2, false>(a, st);
    if (big_env < 8 && (big_env & 4) && n > 128 && n <= 256 && k <= 64) return launch_fused_big<256, BF, 3, 2, true>(a, st);
    if (big_env < 8 && (big_env & 2) && n > 64 && n <= 128) return nsb == 3 ? launch_fused_big<128, BF, 3, 3, true>(a, st) : launch_fused_big<128, BF, 3, 2, true>(a, st);
  }
#endif
  // (round 5: 128-column direct tiles at ANY n -- every column tile re-reading its A rows through L2, as 12544 x 256 x 64 does at 0.89 of its roofline -- measured on
  //  the shapes the A-stationary and wide kernels serve: slower on all of them, 3136 x 512 x 128 x 4 147 vs 129 us, 784 x 1024 x 256 x 6 164 vs 114, 784 x 256 x 2304 x 6
  //  303 vs 195, 196 x 512 x 2048 x 2 55 vs 32 (profiles/ab_direct_any_r05an.txt; the rows whose two columns agree went to the big form before the hook).  Removed.)
  if (!wide_env && (n <= 128 || (n <= 256 && k <= 64))) {
#ifdef SM_TUNING
    if (tuning_int("SM_FUSED_NW", 4) == 8) {  // eight waves of 16 rows per workgroup: the same LDS, twice the waves per SIMD
      if (n <= 64) return launch_fused_direct<64, 2, BF, 128, 8>(a, st);
      return launch_fused_direct<128, 2, BF, 128, 8>(a, st);
    }
    if (tuning_int("SM_FUSED_BM", 128) == 64) {  // 64-row tiles: 32 KiB (n = 64) of LDS per workgroup, five workgroups per CU
      if (n <= 64) return launch_fused_direct<64, 2, BF, 64>(a, st);
      return launch_fused_direct<128, 2, BF, 64>(a, st);
    }
    // (round 5: 192-row tiles at 128 columns -- four waves of 48 rows, a third less B re-read per A byte, 80 KiB: still two workgroups per CU -- bit-identical and
    //  no faster than the 128-row tiles on any n = 128 shape: 3136 x 128 x 1152 x 4 193.5 us against 194, 3136 x 128 x 512 x 3 83.7 against 80, 12544 x 128 x 256 61 against 60
    //  (profiles/ab_bm192_r05s.txt: its "bm128" column is the spilling tuning build, compare with profiles/sweep_r05y_f16_resnet50.txt).  So the B tile re-read through
    //  the CU's L2 port is NOT what holds the 128-column shapes at 4.5 TB/s.  Removed.)
#endif
#ifdef SM_TUNING
    if (tuning_int("SM_DIRECT_NT", 1) == 0) {  // A/B of the non-temporal hint on the A DMA
      if (n <= 64) return launch_fused_direct<64, 2, BF, 128, 4, false>(a, st);
      return launch_fused_direct<128, 2, BF, 128, 4, false>(a, st);
    }
#endif
    if (n <= 64) return direct_env >= 3 ? launch_fused_direct<64, 3, BF>(a, st) : launch_fused_direct<64, 2, BF>(a, st);
    // (round 5) the non-temporal hint on the A loads only where A really is read once and streams long: with two column tiles (n = 256,
    // k = 64) the second tile's read of the same rows then misses L2 -- 12544 x 256 x 64 x 3: 200 -> 166 us without the hint -- and the
    // short-K 128-column layer is 5 % faster without it too (12544 x 128 x 256: 68.7 -> 64.8 us); 3136 x 128 x 512 / 1152 and every
    // 64-column shape keep it (1-6 % faster with it; profiles/nt_ab_r05q.txt).  Same C either way.
    if (direct_env < 3 && (n > 128 || k < 512)) return launch_fused_direct<128, 2, BF, 128, 4, false>(a, st);
    return direct_env >= 3 ? launch_fused_direct<128, 3, BF>(a, st) : launch_fused_direct<128, 2, BF>(a, st);
  }
  // n > 256, short K, plain store: A-stationary (the 2:4 image of a row panel stays in LDS across column tiles)
  static const int astat_env = tuning_int("SM_FUSED_ASTAT", 1);  // tuning aid: 0 = off
  if (astat_env && n > 256 && k <= 512 && beta == 0.0f && aligned16(C) && (strideC % 8 == 0) &&
      astat_lds_bytes((int)(k / 64), 3) <= 160 * 1024) {
    return launch_fused_astat<BF>(a, st);
  }
  // wider: 256-column tiles, 8 consumer waves (wave tile 32 x 128), split loaders; for n <= 256 every row of A is loaded
  // and selected exactly once, beyond that once per 256 columns (callers with n >= 512, a long K and a reusable A are
  // better served by sm_compress24_f16 + sm_spmma_f16: bench.py --path auto decides per layer).  Three A stages in
  // flight, a B ring of 4 and a 2 x 4 consumer grid all measured within noise of this configuration.
  // The persistent form pays off where a tile is short (its fill / drain is a large share of it): 3136 x 256 x 512, b = 32:
  // 39.5 vs 43.9 us per instance, 784 x 256 x 1024: 19.6 vs 20.3; on 36- and 72-stage tiles the statically assigned tiles
  // lose to the hardware's dynamic dispatch (784 x 256 x 2304: 40.5 vs 39.1, 196 x 512 x 4608: 41.0 vs 35.3;
  // profiles/widep_r03g.txt), so those keep one workgroup per tile.  SM_FUSED_WIDEP (tuning aid): 0 = never, 2 = always.
  static const int widep_env = tuning_int("SM_FUSED_WIDEP", 1);
  if (widep_env == 2 || (widep_env == 1 && k <= 1024)) return launch_fused_widep<256, 4, 2, 4, 2, 3, BF>(a, st);
  if (n > 256) return launch_fused_wide<256, 4, 2, 4, 2, 3, BF, false>(a, st);  // several column tiles read the same A rows: no non-temporal hint
  return launch_fused_wide<256, 4, 2, 4, 2, 3, BF>(a, st);
}

extern "C" int sm_spmma_fused_f16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                                  size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                                  sm_stream_t stream) {
  return spmma_fused16<false>(1, &A, &B, &C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream);
}
extern "C" int sm_spmma_fused_bf16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                                   size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                                   sm_stream_t stream) {
  return spmma_fused16<true>(1, &A, &B, &C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream);
}

// The same entry points with a workspace (round 5): the library may then run the stream-K form (spmma_f16_fused_sk_kernel) on the
// shapes whose whole-tile rounds leave CUs idle; every other shape runs exactly as without one.
extern "C" int sm_spmma_fused_workspace_size(size_t* bytes) {
  if (!bytes) {
    set_error("sm_spmma_fused_workspace_size: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  *bytes = sk_workspace_bytes((size_t)device_cu_count(), 256);
  return SM_STATUS_SUCCESS;
}
// State of a stream-K workspace's flag page: 0 = clean (all flags zero: ready for the next launch), 1 = a fix-up timed
// out (word 1023 = 0xdead: a tile of that launch was stored WITHOUT a partial, and the late contributor's flag may still be raised),
// 2 = flags raised but no timeout recorded (a launch is still running, or the page was never zeroed).  Anything but 0 after the
// stream has drained means: discard that launch's C and zero the page (hipMemsetAsync of its first 4096 bytes) before the next call.
// Blocks on `stream` (a 4 KiB read-back): call it outside timed regions.
extern "C" int sm_spmma_fused_workspace_state(const void* workspace, int* state, sm_stream_t stream) {
  if (!workspace || !state) {
    set_error("sm_spmma_fused_workspace_state: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  unsigned h[1024];
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(h, workspace, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return check_launch("sm_spmma_fused_workspace_state");
  bool dirty = false;
  for (int i = 0; i < 1023; ++i) dirty = dirty || h[i] != 0u;
  *state = h[1023] != 0u ? 1 : (dirty ? 2 : 0);
  return SM_STATUS_SUCCESS;
}
// The rule's answer for a launch with beta == 0 and a 16-byte aligned, 8-element-strided C -- what the A-stationary exception of the
// dispatch (spmma_fused16: astat_shape) also asks for; with another beta / C alignment the n > 256, k <= 512 shapes are NOT A-stationary
// shapes and the _ws call may run stream-K where this query says no (state the assumption).
extern "C" int sm_spmma_fused_streamk_plan(size_t rows, size_t n, size_t k, size_t problems, int* takes, unsigned* plan) {
  if (!takes || !plan) {
    set_error("sm_spmma_fused_streamk_plan: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  SkPlan pl = {};
  *takes = sk_takes(rows, problems, n, k, n > 256 && k <= 512, pl) ? 1 : 0;
  unsigned* o = plan;
  *o++ = pl.tg; *o++ = pl.wg; *o++ = pl.groups_full; *o++ = pl.tgl; *o++ = pl.wgl; *o++ = pl.slots; *o++ = pl.units;
  for (int i = 0; i < 9; ++i) *o++ = pl.cut[i];
  for (int i = 0; i < 9; ++i) *o++ = pl.cutl[i];
  return SM_STATUS_SUCCESS;
}
extern "C" int sm_spmma_fused_f16_ws(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                                     size_t strideB, size_t strideC, float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  return spmma_fused16<false>(1, &A, &B, &C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream, workspace, workspace_bytes);
}
extern "C" int sm_spmma_fused_bf16_ws(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                                      size_t strideB, size_t strideC, float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  return spmma_fused16<true>(1, &A, &B, &C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream, workspace, workspace_bytes);
}

// Grouped forms: `count` same-shape problems (host arrays of device pointers, as the reference's batched::spmm takes its
// As / Cs, spmm.hxx:30-33) in as few grids as possible (MAXG problems per launch).  Same kernels, same C bit for bit.
template <bool BF>
static int dense_twin16(const DenseTwinCall& c, hipStream_t st) {
  const size_t n = (size_t)c.N, k = (size_t)c.K;
  const bool tables = c.Ap != nullptr;
  if (tables != (c.Bp != nullptr) || tables != (c.Cp != nullptr)) return SM_STATUS_NOT_SUPPORTED;
  if (c.N % 8 != 0 || c.N < 8 || (!tables && (!aligned16(c.A) || !aligned16(c.B) || !aligned16(c.C)))) return SM_STATUS_NOT_SUPPORTED;
  FusedArgs a = {};
  for (int g = 0; g < MAXG; ++g) { a.A[g] = c.A; a.B[g] = c.B; a.C[g] = c.C; }
  a.dAp = c.Ap; a.dBp = c.Bp; a.dCp = c.Cp;
  a.ngroup = 1;
  a.sA = c.sA; a.sB = c.sB; a.sC = c.sC;
  a.Mrows = c.M; a.N = c.N; a.K = c.K; a.lda = c.lda;
  a.batch = c.batch; a.alpha = c.alpha; a.beta = c.beta;
  // ragged k: the span form -- one tall contiguous A per problem (lda == k), n <= 128, span + whole B inside the LDS
  if (k % 64 != 0 || c.lda % 8 != 0) {
    if (c.lda != c.K || n > 128 || ((size_t)c.M * k * 2) % 16 != 0 || (c.batch > 1 && !tables)) return SM_STATUS_NOT_SUPPORTED;
    if (((size_t)128 * k * 2 + 1152) + ((k + 63) / 64 * 64) * (n <= 64 ? 64 : 128) * 2 > 160 * 1024) return SM_STATUS_NOT_SUPPORTED;
    a.ngroup = c.batch;  // the span kernel indexes problems, not grid batches: with tables every batch entry is a problem
    a.batch = 1;
    if (!tables) a.ngroup = 1;
    return n <= 64 ? launch_fused_span<64, BF, true>(a, st) : launch_fused_span<128, BF, true>(a, st);
  }
  if (c.sA % 8 != 0 || c.sB % 8 != 0) return SM_STATUS_NOT_SUPPORTED;
  // (round 5) with a workspace: the stream-K form -- only on the shapes where this pipeline is the dense GEMM's better one at all
  // (n <= 256 with k >= 2048, below): elsewhere gemm_f16.hip's 128 x 128 tiles at two workgroups per CU beat it by more than a K split
  // returns (196 x 512 x 4608 x 3: 163 us against 110, profiles/ab_streamk_r05e.txt)
  if (c.workspace && c.mode < 2 && n > 128 && n <= 256 && k >= 2048) {
    SkPlan pl;
    if (sk_takes((size_t)c.M, (size_t)c.batch, n, k, false, pl) && c.workspace_bytes >= sk_workspace_bytes((size_t)pl.slots * ((n + 255) / 256), 256) &&
        aligned16(c.workspace))
      return n <= 256 ? launch_fused_sk<256, BF, true, true>(a, c.workspace, c.workspace_bytes, st) : launch_fused_sk<256, BF, false, true>(a, c.workspace, c.workspace_bytes, st);
  }
  if (n <= 128) {  // gemm_f16.hip's 128 x 64 / 128 x 128 tiles are the direct pipeline already (A/B in tuning builds only)
#ifdef SM_TUNING
    if (c.mode >= 2) return n <= 64 ? launch_fused_direct<64, 2, BF, 128, 4, true, true>(a, st) : launch_fused_direct<128, 2, BF, 128, 4, true, true>(a, st);
#endif
    return SM_STATUS_NOT_SUPPORTED;
  }
  if (k <= 64) return SM_STATUS_NOT_SUPPORTED;
  if (c.mode < 2 && ((size_t)c.M + 255) / 256 * 256 * 100 > (size_t)c.M * 115) return SM_STATUS_NOT_SUPPORTED;  // > 15 % of the 256-row tiles would be padding
  // n > 128: 256 x 256 tiles (A streamed once per 256 columns instead of once per 128).  Measured per ResNet-50 shape against
  // gemm_f16.hip's 128 x 128 tiles, two workgroups per CU (profiles/ab_dense_r04r2.txt): they win only on the long-K 256-column
  // shape (784 x 256 x 2304 x 6: 40.1 -> 35.6 us per instance) and lose 5-17 % on 3136 x 256 x 512, 3136 x 512 x 128,
  // 196 x 2048 x 512, 196 x 512 x 4608 -- the rule is that one case.
  if (c.mode < 2 && !(n <= 256 && k >= 2048)) return SM_STATUS_NOT_SUPPORTED;
  return n <= 256 ? launch_fused_big<256, BF, 3, 2, true, true>(a, st) : launch_fused_big<256, BF, 3, 2, false, true>(a, st);
}

int sm::gemm_dense_twin(const DenseTwinCall& c, hipStream_t st) { return c.bf ? dense_twin16<true>(c, st) : dense_twin16<false>(c, st); }

template <bool BF>
static int spmma_fused16_grouped(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n, size_t k,
                                 size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                                 sm_stream_t stream, void* workspace = nullptr, size_t workspace_bytes = 0) {
  if (count && (!A || !B || !C)) {
    set_error("sm_spmma_fused_{f16,bf16}_grouped: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  for (size_t i = 0; i < count; i += (size_t)MAXG) {
    const size_t ng = count - i < (size_t)MAXG ? count - i : (size_t)MAXG;
    // (the launches of one call follow one another on the stream: they can share the workspace)
    if (const int rc = spmma_fused16<BF>(ng, A + i, B + i, C + i, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream, workspace, workspace_bytes)) return rc;
  }
  return SM_STATUS_SUCCESS;
}
extern "C" int sm_spmma_fused_f16_grouped(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n,
                                          size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                                          float alpha, float beta, sm_stream_t stream) {
  return spmma_fused16_grouped<false>(count, A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream);
}
extern "C" int sm_spmma_fused_bf16_grouped(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n,
                                           size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                                           float alpha, float beta, sm_stream_t stream) {
  return spmma_fused16_grouped<true>(count, A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream);
}
extern "C" int sm_spmma_fused_f16_grouped_ws(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n,
                                             size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                                             float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  return spmma_fused16_grouped<false>(count, A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream, workspace, workspace_bytes);
}
extern "C" int sm_spmma_fused_bf16_grouped_ws(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n,
                                              size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                                              float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  return spmma_fused16_grouped<true>(count, A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream, workspace, workspace_bytes);
}
