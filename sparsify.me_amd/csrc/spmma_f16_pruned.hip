// spmma_f16_pruned.hip -- the call sequence of sparsifyme::spmma() as ONE kernel (round 4): magnitude-prune A to 2:4 IN
// PLACE (TILE rule as the reference asks for, or STRIP), raise the validity flag, and multiply -- reference
// include/sparsify.me/spmma.hxx:82-113: cusparseLtSpMMAPrune(dA -> dA, TILE), cusparseLtSpMMAPruneCheck, Compress, Matmul.
// No compressed blob exists: per call A is read once and written once (the pruned operand the reference leaves in dA),
// instead of read + written + blob written (one-pass prune kernel) + blob read (matmul): A + A + B + C bytes instead of
// 3.1 A + B + C.
//
// Structure = the direct fused kernel (spmma_f16_fused.hip: dense A stage and B stage by LDS-DMA, ring of 2, four waves,
// wave w owns rows 32 w .. 32 w + 31 and all columns) with a PRUNE PHASE between a stage's arrival and its use: thread
// (q, c) = (tid / 8, tid % 8) takes rows 4 q .. 4 q + 3 x the 16-byte chunk c of the stage's dense image in LDS -- two
// 4 x 4 tiles, the item of prune_fused.hip -- applies the frozen rule (select24.h: tile_select_pairs / strip_select_f16),
// writes the pruned rows back into the LDS image AND to A in global memory (eight whole 128-byte lines per wave store),
// and after one more barrier the stage is consumed exactly as the direct kernel consumes a dense stage: the feeding lane's
// STRIP selection of the PRUNED strip is what sm_compress24 stores for it, so C is bit-identical to
// sm_spmma(sm_compress24(sm_prune24(A))) and A to sm_prune24(A) (tests/test_gpu_parity.py::test_prune_spmma_one_kernel).
// The flag is derived from the values about to be stored (more than two of a strip != 0), as sm_prune24_check derives it
// from the stored matrix.  n <= 128 (one column tile per row panel: every element of A has exactly one reader and writer),
// k % 64 == 0, m % 4 == 0; everything else: SM_STATUS_NOT_SUPPORTED and the caller runs sm_prune24_compress24 + sm_spmma.
#include "select24.h"
#include "spmma_args.h"

namespace sm {

struct PrunedArgs {
  const half_t* Ain;
  half_t* A;  // the pruned operand: Ain itself (in place, what the reference does) or a second buffer
  const half_t* B;
  half_t* C;
  int* d_valid;
  size_t sA, sB, sC;
  int Mrows, N, K, lda;
  int batch, tiles_m;
  float alpha, beta;
};

template <int BN, bool BF, bool TILE>
__global__ __launch_bounds__(256) void spmma_f16_pruned_kernel(const PrunedArgs p) {
  constexpr int BM = 128, NW = 4, TM = BM / NW, FM = TM / 16, FN = BN / 16;  // ring of 2 stage buffers
  constexpr int SA = BM * 128, SB = 64 * BN * 2, STAGE = SA + SB;
  constexpr int A_N = BM / 8, B_N = BN / 8, W = A_N + B_N, SL = W / NW;
  static_assert(W % NW == 0 && A_N % NW == 0, "equal DMA share per wave; A / B split per instruction index");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lid = tile_order(blockIdx.x, gridDim.x, true);  // one column tile per row panel: dispatch order (mma_tile.h)
  const unsigned b = lid / (unsigned)p.tiles_m, tile_m = lid - b * (unsigned)p.tiles_m;
  const int m0 = (int)tile_m * BM;
  const int nkt = p.K / 64;
  const half_t* Ain = p.Ain + (size_t)b * p.sA;
  half_t* A = p.A + (size_t)b * p.sA;
  const half_t* B = p.B + (size_t)b * p.sB;
  half_t* C = p.C + (size_t)b * p.sC;
  const int mlast = p.Mrows - 1;

  const char* src[SL];
  size_t step[SL];
  unsigned loff[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    if (t < (unsigned)A_N) {
      const unsigned row = 8u * t + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      src[i] = reinterpret_cast<const char*>(Ain + (size_t)gr * p.lda) + 16u * cs;
      step[i] = 128;
      loff[i] = t * 1024u;
    } else {
      const unsigned j = t - A_N, panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ b_swz(kr);
      int gc = (int)(64u * panel + 8u * cs);
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      src[i] = reinterpret_cast<const char*>(B + (size_t)kr * p.N + gc);
      step[i] = (size_t)64 * p.N * 2;
      loff[i] = SA + panel * 8192u + (j & 7u) * 1024u;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
    // A is read exactly once by the whole grid (one column tile per row panel): non-temporal, as in the direct fused kernel
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      if (NW * i < A_N) __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)kt * step[i]), (lptr_t*)(base + loff[i]), 16, 0, 2);
      else __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)kt * step[i]), (lptr_t*)(base + loff[i]), 16, 0, 0);
    }
  };

  // prune phase geometry: rows 4 q + i (i < 4), chunk c of the stage
  const unsigned q = tid >> 3, c = tid & 7u;
  unsigned poff[4];
  half_t* gdst[4];
  bool gok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = 4u * q + i;
    poff[i] = a_off(row, c);
    const int gr = m0 + (int)row;
    gok[i] = gr <= mlast;
    gdst[i] = A + (size_t)(gok[i] ? gr : mlast) * p.lda + 8u * c;
  }
  bool bad = false;

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  if (nkt > 0) stage(0, 0);
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    // stage kt has landed for every wave -- and this wave's stores of stage kt - 1's pruned rows have left (vmcnt counts them
    // with the DMA, in order) -- and every wave has left the buffer about to be refilled
    wait_dma_and_barrier<0>();
    if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);
    char* As = smem + cur * STAGE;
    // ---- prune phase: two 4 x 4 tiles (or eight strips) per thread, in place in LDS and out to global A
    {
      u4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u4*>(As + poff[i]);
      uint32_t o[4][4];
      if constexpr (TILE) {
#pragma unroll
        for (unsigned t = 0; t < 2; ++t) {
          float mag[4][4];
#pragma unroll
          for (unsigned r = 0; r < 4; ++r) {
            mag2<BF>(v[r][2 * t], mag[r][0], mag[r][1]);
            mag2<BF>(v[r][2 * t + 1], mag[r][2], mag[r][3]);
          }
          unsigned top, bot;
          tile_select_pairs(mag, top, bot);
          const unsigned pr[4] = {top >> 3, top & 7u, bot >> 3, bot & 7u};
#pragma unroll
          for (unsigned r = 0; r < 4; ++r) strip_mask(v[r][2 * t], v[r][2 * t + 1], pair_rowmask(pr[r]), o[r][2 * t], o[r][2 * t + 1]);
          __builtin_amdgcn_sched_barrier(0);  // one tile after the other (registers)
        }
      } else {
#pragma unroll
        for (unsigned r = 0; r < 4; ++r)
#pragma unroll
          for (unsigned t = 0; t < 2; ++t) {
            uint32_t kp, nb;
            strip_select_f16(v[r][2 * t], v[r][2 * t + 1], kp, nb);
            strip_mask(v[r][2 * t], v[r][2 * t + 1], (1u << (nb & 3u)) | (1u << (nb >> 2)), o[r][2 * t], o[r][2 * t + 1]);
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u4 w = u4{o[i][0], o[i][1], o[i][2], o[i][3]};
        *reinterpret_cast<u4*>(As + poff[i]) = w;
        if (gok[i]) __builtin_nontemporal_store(w, reinterpret_cast<u4*>(gdst[i] + (size_t)kt * 64));  // written once, not read again here
        // the flag comes from what is stored: a strip (two dwords) with more than two halves != 0 (-0 counts as zero)
#pragma unroll
        for (unsigned t = 0; t < 2; ++t) {
          const uint32_t x = o[i][2 * t] & 0x7fff7fffu, y = o[i][2 * t + 1] & 0x7fff7fffu;
          const unsigned nz = ((x & 0xffffu) != 0u) + ((x >> 16) != 0u) + ((y & 0xffffu) != 0u) + ((y >> 16) != 0u);
          bad |= nz > 2u;
        }
      }
    }
    // the pruned image is complete for every wave (LDS writes only: the DMA of stage kt + 1 and the stores stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    smfmac_stage_dense_a<FM, FN, BF>(As, As + SA, wave * TM, 0, lane, acc);
    cur ^= 1;
  }
  if (p.d_valid && __any(bad)) {
    if (lane == 0) raise_flag(p.d_valid);
  }
  __syncthreads();
  store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, true, wave * TM, 0, m0, 0, p.Mrows, p.N, p.alpha, p.beta, tid);
}

template <int BN, bool BF, bool TILE>
static int launch_pruned(const PrunedArgs& a0, hipStream_t st) {
  PrunedArgs a = a0;
  a.tiles_m = (a.Mrows + 127) / 128;
  const size_t nwg = (size_t)a.tiles_m * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_prune24_spmma: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t stage_bytes = 128 * 128 + 64 * BN * 2;
  constexpr size_t lds_epi = (size_t)128 * (BN * 2 + 16);
  const size_t lds_main = (a.K / 64 < 2 ? 1 : 2) * stage_bytes;
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  constexpr size_t lds_max = 2 * stage_bytes > lds_epi ? 2 * stage_bytes : lds_epi;
  static LdsOptIn lds_optin;
  if (lds_max > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f16_pruned_kernel<BN, BF, TILE>), lds_max, "spmma_f16_pruned_kernel")) return rc;
  }
  spmma_f16_pruned_kernel<BN, BF, TILE><<<dim3((unsigned)nwg), dim3(256), lds, st>>>(a);
  return check_launch("spmma_f16_pruned_kernel");
}

template <bool BF>
static int prune24_spmma16(const void* A_in, void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                           size_t strideB, size_t strideC, int alg, int* d_valid, float alpha, float beta, sm_stream_t stream) {
  if (!A_in || !A || !B || !C || lda < k || (alg != 0 && alg != 1)) {
    set_error("sm_prune24_spmma_{f16,bf16}: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (n > 128 || n % 8 != 0 || k == 0 || k % 64 != 0 || m % 4 != 0 || lda % 8 != 0 || strideA % 8 != 0 || strideB % 8 != 0 || !aligned16(A) || !aligned16(A_in) || !aligned16(B) ||
      m * batch > 0x7fffffffull || k > 0x7fffffffull || lda > 0x7fffffffull) {
    // (round 6) Every other shape the exact fused kernels take (n > 128: 35 of ResNet-50's 49 layers; ragged k: the stem layer): TWO launches and
    // still no blob -- (1) prune in place + flag as one pass over A (sm_prune24_compress24_* with a null blob: read A, write A), (2) the fused
    // kernel on the PRUNED operand: its STRIP selection of a 2:4 strip is what sm_compress24 stores for that strip, so C is the staged
    // sequence's bit for bit.  HBM bytes: 2 A + C + B when the pruned A (<= 116 MB on those layers) is still in the 256 MiB Infinity Cache
    // for launch 2, against 3.125 A + C + B of the blob pair.  Why not ONE kernel there: with several column tiles per row panel every
    // element of A has several readers and an in-place writer (a torn 4 x 4 tile prunes differently), and the few-tile shapes (98-196
    // workgroups) would run the TILE rule -- ~520 VALU instructions per tile, the API path's real bound -- on a fraction of
    // the chip's SIMDs; the one-pass prune spreads it over all of them.
    if (k == 0 || !spmma_fused16_takes_exact(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC)) {
      set_error("sm_prune24_spmma_{f16,bf16}: shape / alignment not taken by the one-kernel form nor by the exact fused kernels "
                "(use sm_prune24_compress24 + sm_spmma)");
      return SM_STATUS_NOT_SUPPORTED;
    }
    int rc = BF ? sm_prune24_compress24_bf16(A_in, A, m, k, lda, batch, strideA, nullptr, d_valid, alg, stream)
                : sm_prune24_compress24_f16(A_in, A, m, k, lda, batch, strideA, nullptr, d_valid, alg, stream);
    if (rc != SM_STATUS_SUCCESS) return rc;
    return BF ? sm_spmma_fused_bf16(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream)
              : sm_spmma_fused_f16(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream);
  }
  PrunedArgs a = {};
  a.Ain = (const half_t*)A_in; a.A = (half_t*)A; a.B = (const half_t*)B; a.C = (half_t*)C; a.d_valid = d_valid;
  a.sA = strideA; a.sB = strideB; a.sC = strideC;
  a.Mrows = (int)m; a.N = (int)n; a.K = (int)k; a.lda = (int)lda; a.batch = (int)batch;
  a.alpha = alpha; a.beta = beta;
  if (batch > 1 && strideB == 0 && strideA == m * lda && strideC == m * n) {  // one tall matrix (m % 4 == 0: no tile spans two batches)
    a.Mrows = (int)(m * batch);
    a.batch = 1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (d_valid && hipMemsetAsync(d_valid, 0, sizeof(int), st) != hipSuccess) return check_launch("hipMemsetAsync(d_valid)");
  // (a 256-column instantiation -- one workgroup per CU, 128 accumulator registers per lane -- was measured for 128 < n <= 256 and
  // is slower than the two-launch pair on every such ResNet-50 shape, profiles/api_path_r04o.txt: 784 x 256 x 2304 125 vs 109 us,
  // 12544 x 256 x 64 112 vs 75 us; not instantiated)
  if (alg == 0) return n <= 64 ? launch_pruned<64, BF, true>(a, st) : launch_pruned<128, BF, true>(a, st);
  return n <= 64 ? launch_pruned<64, BF, false>(a, st) : launch_pruned<128, BF, false>(a, st);
}

}  // namespace sm

extern "C" int sm_prune24_spmma_f16(const void* A_in, void* A_out, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                                    size_t strideB, size_t strideC, int alg, int* d_valid, float alpha, float beta, sm_stream_t stream) {
  return sm::prune24_spmma16<false>(A_in, A_out, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alg, d_valid, alpha, beta, stream);
}
extern "C" int sm_prune24_spmma_bf16(const void* A_in, void* A_out, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                                     size_t strideB, size_t strideC, int alg, int* d_valid, float alpha, float beta, sm_stream_t stream) {
  return sm::prune24_spmma16<true>(A_in, A_out, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alg, d_valid, alpha, beta, stream);
}
