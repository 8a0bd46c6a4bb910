// gemm_f32.hip -- fp32 matrix kernels on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32
// (result bit-for-bit a k-ordered fmaf chain; gfx950 has no TF32/xf32 path):
//   sm_gemm_rowmajor_f32 / sm_gemm_batched_f32  -- dense, replaces cublasSgemmBatched
//                                                  (include/sparsify.me/gemm.hxx:133-134)
//   sm_spmma_f32                                 -- 2:4 A from the compressed blob; there is no fp32
//                                                  sparse matrix instruction, so the kept values are
//                                                  expanded into the dense LDS tile while staging and
//                                                  the contraction runs dense: A's HBM bytes halve,
//                                                  the MFMA work does not
//   sm_gemm_batched_f64                          -- fp64, plain FMA tiles (lowest priority, a5)
// One register-staged, LDS-tiled kernel template serves the fp32 variants: 128 x BN tile, BK = 32,
// 256 threads = 4 waves; the operands of the MFMA are swapped so a lane ends with four consecutive
// columns of one row and stores them as one 16-byte access.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mma_tile.h"

namespace sm {

struct Gemm32Args {
  const float* A;      // dense A (row-major M x K, lda) -- or null when `vals` is set
  const char* vals;    // 2:4 blob values, stage-major [kc/64][Mtot][32] floats
  const char* meta;    // 2:4 blob metadata, stage-major [kc/64][Mtot][8 B]
  size_t Mtot;         // rows of the whole blob (m * batch)
  const float* B;
  float* C;
  const float* const* Ap;
  const float* const* Bp;
  float* const* Cp;
  size_t sA, sB, sC;   // batch strides (elements); for the blob sA counts ROWS (m)
  int M, N, K, kc;
  int lda, ldb, ldc;
  int batch, tiles_m, tiles_n;
  float alpha, beta;
};

constexpr int BK32 = 32;
constexpr int APITCH = BK32 + 1;  // floats per A row in LDS: column reads by 16 rows x 2 k hit 32 distinct banks

// MODE 0: dense A, B row-major [k][n];  MODE 1: A from the 2:4 blob;  MODE 2: dense A, B given K-MAJOR
// ([n][k], ldb = row pitch in k): the form the Blocked-ELL path needs (its expanded A is row-major [m][k]
// and plays the B role of the transposed product).
// ATR (dense modes only): A given M-CONTIGUOUS (element (r, kk) at A[kk * lda + r]) -- with MODE 2 the two transposed
// operand forms of sm_gemm_batched_f32.
template <int BM, int BN, int WM, int WN, int MODE, bool ATR = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const Gemm32Args p) {
  constexpr bool SPARSE = MODE == 1, BKM = MODE == 2;
  static_assert(!(ATR && SPARSE), "the blob is row-major by construction");
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int BPITCH = BN + 16;  // the two k-rows a 32-lane half reads land on different bank halves
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);     // [BM][APITCH]
  float* Bs = As + BM * APITCH;                   // [BK32][BPITCH]

  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned wm = wave / WN, wn = wave % WN;
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = lid / tiles, trem = lid - b * tiles;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;

  const float* A = SPARSE ? nullptr : (p.Ap ? p.Ap[b] : p.A + (size_t)b * p.sA);
  const float* B = p.Bp ? p.Bp[b] : p.B + (size_t)b * p.sB;
  float* C = p.Cp ? p.Cp[b] : p.C + (size_t)b * p.sC;
  const size_t row_base = SPARSE ? (size_t)b * p.sA : 0;  // first blob row of this grid batch
  const bool a_vec = !SPARSE && (p.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15u) == 0);
  const bool b_vec = (p.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15u) == 0);

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  // staging registers: A tile BM x 32 floats = BM*8 float4 / 256 threads; B tile 32 x BN
  constexpr int A_CH = BM * 8 / 256, B_CH = 32 * (BN / 4) / 256;
  f4 ra[A_CH], rb[B_CH];

  auto load4 = [](const float* rowp, int col, int limit, bool row_ok, bool vec) -> f4 {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (!row_ok || col >= limit) return v;
    if (vec && col + 4 <= limit) return *reinterpret_cast<const f4*>(rowp + col);
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = (col + t < limit) ? rowp[col + t] : 0.0f;
    return v;
  };

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const unsigned q = tid + 256u * i, row = q >> 3, ch = q & 7u;  // 8 strips of 4 dense k per row
      const int gr = m0 + (int)row, kk = k0 + 4 * (int)ch;
      if constexpr (ATR) {
        const unsigned kr = q / (BM / 4), rc = q % (BM / 4);  // 4 consecutive rows of one k
        const int gk = k0 + (int)kr;
        ra[i] = load4(A + (size_t)gk * p.lda, m0 + 4 * (int)rc, p.M, gk < p.K, a_vec);
      } else if constexpr (SPARSE) {
        // expand one strip: two kept values + their nibble -> four dense k
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (gr < p.M && kk < p.kc) {
          const size_t R = row_base + (size_t)gr;
          const float* vp = reinterpret_cast<const float*>(p.vals) + ((size_t)(kk >> 6) * p.Mtot + R) * 32 + ((kk & 63) >> 1);
          const float a0 = vp[0], a1 = vp[1];
          const unsigned mb = *reinterpret_cast<const unsigned char*>(p.meta + ((size_t)(kk >> 6) * p.Mtot + R) * 8 + ((kk >> 3) & 7));
          const unsigned nib = (mb >> (4 * ((kk >> 2) & 1))) & 0xfu, p0 = nib & 3u, p1 = nib >> 2;
#pragma unroll
          for (unsigned t = 0; t < 4; ++t) v[t] = t == p0 ? a0 : (t == p1 ? a1 : 0.0f);
        }
        ra[i] = v;
      } else {
        ra[i] = load4(A + (size_t)gr * p.lda, kk, p.K, gr < p.M, a_vec);
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      if constexpr (BKM) {
        const unsigned q = tid + 256u * i, nrow = q >> 3, ch = q & 7u;
        const int gn = n0 + (int)nrow;
        rb[i] = load4(B + (size_t)gn * p.ldb, k0 + 4 * (int)ch, p.K, gn < p.N, b_vec);
      } else {
        const unsigned q = tid + 256u * i, kr = q / (BN / 4), cn = q % (BN / 4);
        const int gk = k0 + (int)kr;
        rb[i] = load4(B + (size_t)gk * p.ldb, n0 + 4 * (int)cn, p.N, gk < p.K, b_vec);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const unsigned q = tid + 256u * i, row = q >> 3, ch = q & 7u;
      if constexpr (ATR) {
        const unsigned kr = q / (BM / 4), rc = q % (BM / 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) As[(4 * rc + t) * APITCH + kr] = ra[i][t];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) As[row * APITCH + 4 * ch + t] = ra[i][t];
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      if constexpr (BKM) {
        const unsigned q = tid + 256u * i, nrow = q >> 3, ch = q & 7u;
#pragma unroll
        for (int t = 0; t < 4; ++t) Bs[nrow * APITCH + 4 * ch + t] = rb[i][t];
      } else {
        const unsigned q = tid + 256u * i, kr = q / (BN / 4), cn = q % (BN / 4);
        *reinterpret_cast<f4*>(Bs + kr * BPITCH + 4 * cn) = rb[i];
      }
    }
  };

  const int Kloop = SPARSE ? p.kc : p.K;
  const int nkt = (Kloop + BK32 - 1) / BK32;
  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < nkt) gload((kt + 1) * BK32);
#pragma unroll
    for (int s = 0; s < BK32 / 4; ++s) {
      float af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = As[(wm * TM + i * 16 + (lane & 15u)) * APITCH + 4 * s + (lane >> 4)];
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bf[j] = BKM ? Bs[(wn * TN + j * 16 + (lane & 15u)) * APITCH + 4 * s + (lane >> 4)]
                    : Bs[(4 * s + (lane >> 4)) * BPITCH + wn * TN + j * 16 + (lane & 15u)];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // swapped operands: lane holds C[row lane&15][cols 4*(lane>>4) .. +3]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  const bool c_vec = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int gr = m0 + (int)(wm * TM + i * 16 + (lane & 15u));
      const int gc = n0 + (int)(wn * TN + j * 16 + 4u * (lane >> 4));
      if (gr >= p.M || gc >= p.N) continue;
      float* dst = C + (size_t)gr * p.ldc + gc;
      f4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = p.alpha * acc[i][j][r];
      if (c_vec && gc + 4 <= p.N) {
        if (p.beta != 0.0f) {
          const f4 old = *reinterpret_cast<const f4*>(dst);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p.beta * old[r];
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));  // C is written once: keep it out of the operands' way in L2
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (gc + r < p.N) dst[r] = p.beta != 0.0f ? v[r] + p.beta * dst[r] : v[r];
      }
    }
}

template <int BM, int BN, int WM, int WN, int MODE, bool ATR = false>
static int launch32(const Gemm32Args& a0, hipStream_t st) {
  Gemm32Args a = a0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("gemm_f32: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_b = (size_t)BK32 * (BN + 16) > (size_t)BN * APITCH ? (size_t)BK32 * (BN + 16) : (size_t)BN * APITCH;
  constexpr size_t lds = ((size_t)BM * APITCH + lds_b) * sizeof(float);
  gemm_f32_kernel<BM, BN, WM, WN, MODE, ATR><<<dim3((unsigned)nwg), dim3(256), lds, st>>>(a);
  return check_launch("gemm_f32_kernel");
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA pipeline for the dense fp32 product (K % 32 == 0, 16-byte aligned rows, N % 4 == 0): the structure of the
// fp16 kernels (gemm_f16.hip) in fp32 bytes.  Stage = 32 k: A [BM][128 B] (16-byte chunk c of row r at chunk
// c ^ (r & 7), a_off) and B in panels of 32 columns [32 k][128 B] (chunk c of k-row kr at c ^ 4 * ((kr >> 3) & 1)),
// all by global_load_lds into a ring of NS stage buffers, counted vmcnt + one barrier per stage.  The four k-slots of
// v_mfma_f32_16x16x4_f32 are fed k = 8 g + s at step s (g = lane / 16) instead of 4 s + g: a lane then owns 8
// CONSECUTIVE k of its A row per stage (two ds_read_b128 instead of eight ds_read_b32); B is read one float per step
// (the two lane groups of a 32-lane half, k-rows 8 apart, sit in different halves of the 128-byte row).  Summation
// order within a stage differs from the register-staged kernel's (still one fp32 fma chain per output).
// BKM: B given k-major ([n][k], the Blocked-ELL path): its image is an A-style image and its reads are b128 too.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned b32_off(unsigned kr, unsigned col) {  // byte offset in the [panel][32][128 B] image
  return (col >> 5) * 4096u + kr * 128u + 16u * (((col & 31u) >> 2) ^ (4u * ((kr >> 3) & 1u))) + 4u * (col & 3u);
}

// P24 (sm_spmma_fused_f32): the 2:4 STRIP rule is applied to each stage's A image in LDS, in place, before the MFMA reads
// it -- so C = prune24_strip(A) * B without a blob and without a compress pass.  fp32 has no sparse matrix instruction: the matrix work is the dense
// kernel's, what is saved is the 2.06 ms compress pass and the blob round trip of the staged pair (ResNet-18 table).
// P24 = 2 (round 4): the same rule in the REGISTERS of the lane that feeds the strip to the MFMA -- a lane's two ds_read_b128 of
// a stage are two whole strips of its row -- for tilings with ONE wave column (WN == 1: a wave owns its rows across all columns
// of the tile, so every strip is selected exactly once; with WN > 1 the selection would repeat per wave column, which is what
// made round 2's register form lose).  No pass over the LDS image, no extra barrier; the ~50 VALU per stage sit in the shadow
// of the stage's 64-128 MFMAs.  Same values reach the MFMA: bit-identical C.
template <int BM, int BN, int WM, int WN, int NS, bool BKM, int P24 = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_dma_kernel(const Gemm32Args p) {
  static_assert(P24 != 2 || WN == 1, "register form of the 2:4 rule: one wave column");
  constexpr int NW = WM * WN, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int SA = BM * 128, SB = BN * 128, STAGE = SA + SB;
  constexpr int A_N = BM / 8, B_N = BN / 8, W = A_N + B_N;  // 1 KiB DMA wave-instructions per stage
  static_assert(W % NW == 0, "equal DMA share per wave");
  constexpr int SL = W / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned wm = wave / WN, wn = wave % WN;
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = lid / tiles, trem = lid - b * tiles;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const float* A = p.Ap ? p.Ap[b] : p.A + (size_t)b * p.sA;
  const float* B = p.Bp ? p.Bp[b] : p.B + (size_t)b * p.sB;
  float* C = p.Cp ? p.Cp[b] : p.C + (size_t)b * p.sC;
  const int mlast = p.M - 1, nlast = p.N - 1;

  const char* src[SL];
  size_t step[SL];
  unsigned loff[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    if (t < (unsigned)A_N) {  // 8 rows x 128 B
      const unsigned row = 8u * t + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      src[i] = reinterpret_cast<const char*>(A + (size_t)gr * p.lda + 4u * cs);
      step[i] = 128;
      loff[i] = t * 1024u;
    } else if constexpr (BKM) {  // B rows are output columns, k-contiguous: the same image as A
      const unsigned j = t - A_N, row = 8u * j + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
      int gn = n0 + (int)row;
      gn = gn < nlast ? gn : nlast;
      src[i] = reinterpret_cast<const char*>(B + (size_t)gn * p.ldb + 4u * cs);
      step[i] = 128;
      loff[i] = SA + j * 1024u;
    } else {  // 8 k-rows of one 32-column panel
      const unsigned j = t - A_N, panel = j >> 2, kr = 8u * (j & 3u) + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ (4u * ((kr >> 3) & 1u));
      int gc = n0 + (int)(32u * panel + 4u * cs);
      gc = gc <= p.N - 4 ? gc : p.N - 4;
      src[i] = reinterpret_cast<const char*>(B + (size_t)kr * p.ldb + gc);
      step[i] = (size_t)32 * p.ldb * 4;
      loff[i] = SA + panel * 4096u + (j & 3u) * 1024u;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < SL; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)kt * step[i]), (lptr_t*)(base + loff[i]), 16, 0, 0);
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / 32;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) stage(s, s);
  const unsigned g = lane >> 4, r = lane & 15u;
  int cur = 0, fill = NS - 1;
  for (int kt = 0; kt < nkt; ++kt) {
    const int ahead = (nkt - 1 - kt) < (NS - 2) ? (nkt - 1 - kt) : (NS - 2);
    if (NS >= 3 && ahead == 1) wait_dma_and_barrier<SL>();
    else wait_dma_and_barrier<0>();
    if (kt + NS - 1 < nkt) stage(kt + NS - 1, fill);
    const char* As = smem + cur * STAGE;
    const char* Bs = As + SA;
    if constexpr (P24 == 1) {
      // the stage's A image pruned in place, once, by all threads: every 16-byte chunk of the image is one strip (4
      // consecutive k of a row, wherever the swizzle put it).  In the MFMA-feeding lanes' registers the same selection
      // ran once per wave column and cost 40 % over the dense kernel (profiles/bench_f32_r02m_regmask.json).
      char* Aw = smem + cur * STAGE;
      for (unsigned q = tid; q < (unsigned)(BM * 8); q += 64u * NW) {
        f4 v = *reinterpret_cast<const f4*>(Aw + q * 16u);
        const u4 kk = __builtin_bit_cast(u4, v);
        const unsigned mk = strip_keepmask(key_of(kk[0]), key_of(kk[1]), key_of(kk[2]), key_of(kk[3]));
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (mk >> t) & 1u ? v[t] : 0.0f;
        *reinterpret_cast<f4*>(Aw + q * 16u) = v;
      }
      __syncthreads();
    }
    f4 alo[FM], ahi[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const unsigned row = wm * TM + i * 16 + r;
      alo[i] = *reinterpret_cast<const f4*>(As + a_off(row, 2u * g));
      ahi[i] = *reinterpret_cast<const f4*>(As + a_off(row, 2u * g + 1u));
      if constexpr (P24 == 2) {
        const u4 kl = __builtin_bit_cast(u4, alo[i]), kh = __builtin_bit_cast(u4, ahi[i]);
        const unsigned ml = strip_keepmask(key_of(kl[0]), key_of(kl[1]), key_of(kl[2]), key_of(kl[3]));
        const unsigned mh = strip_keepmask(key_of(kh[0]), key_of(kh[1]), key_of(kh[2]), key_of(kh[3]));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          alo[i][t] = (ml >> t) & 1u ? alo[i][t] : 0.0f;
          ahi[i][t] = (mh >> t) & 1u ? ahi[i][t] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float bv[8];
      const unsigned col = wn * TN + j * 16 + r;
      if constexpr (BKM) {
        const f4 lo = *reinterpret_cast<const f4*>(Bs + a_off(col, 2u * g));
        const f4 hi = *reinterpret_cast<const f4*>(Bs + a_off(col, 2u * g + 1u));
#pragma unroll
        for (int s = 0; s < 4; ++s) { bv[s] = lo[s]; bv[4 + s] = hi[s]; }
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) bv[s] = *reinterpret_cast<const float*>(Bs + b32_off(8u * g + s, col));
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < FM; ++i)
          // swapped operands: lane holds C[row lane&15][cols 4*(lane>>4) .. +3]; k-slot g carries k = 8 g + s
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[s], s < 4 ? alo[i][s] : ahi[i][s - 4], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
    fill = fill + 1 == NS ? 0 : fill + 1;
  }

  const bool c_vec = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int gr = m0 + (int)(wm * TM + i * 16 + r);
      const int gc = n0 + (int)(wn * TN + j * 16 + 4u * g);
      if (gr >= p.M || gc >= p.N) continue;
      float* dst = C + (size_t)gr * p.ldc + gc;
      f4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = p.alpha * acc[i][j][q];
      if (c_vec && gc + 4 <= p.N) {
        if (p.beta != 0.0f) {
          const f4 old = *reinterpret_cast<const f4*>(dst);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += p.beta * old[q];
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (gc + q < p.N) dst[q] = p.beta != 0.0f ? v[q] + p.beta * dst[q] : v[q];
      }
    }
}

template <int BM, int BN, int WM, int WN, int NS, bool BKM, int P24 = 0>
static int launch32_dma(const Gemm32Args& a0, hipStream_t st) {
  Gemm32Args a = a0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("gemm_f32: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds = (size_t)NS * (BM + BN) * 128;
  static LdsOptIn lds_optin;
  if (lds > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&gemm_f32_dma_kernel<BM, BN, WM, WN, NS, BKM, P24>), lds, "gemm_f32_dma_kernel")) return rc;
  }
  gemm_f32_dma_kernel<BM, BN, WM, WN, NS, BKM, P24><<<dim3((unsigned)nwg), dim3(64 * WM * WN), lds, st>>>(a);
  return check_launch("gemm_f32_dma_kernel");
}

// ---------------------------------------------------------------------------------------------
// 2:4 fp32 product on the same pipeline (K % 64 == 0, even row counts, N % 4 == 0).  Stage = one 64-k plane of the
// blob: the plane's values are 128 B per row -- exactly the dense kernel's A image -- plus 8 B of metadata per row
// (one 1 KiB LDS-DMA instruction for the tile's 128 rows) and B for 64 k (two 32-k images).  There is no fp32 sparse
// matrix instruction, so a lane expands its strips in registers on the way to the MFMA: for half h of the stage it
// owns dense k 32 h + 8 g .. + 7 = two strips = ONE 16-byte chunk of kept values (chunk 4 h + g) and ONE metadata
// byte (4 h + g); A's HBM bytes are 9/16 of the dense kernel's, the matrix work is the same.
// ---------------------------------------------------------------------------------------------
template <int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void spmma_f32_dma_kernel(const Gemm32Args p) {
  constexpr int BM = 128, NW = WM * WN, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;  // ring of 2
  constexpr int SA = BM * 128, SM_ = BM * 8, SBH = BN * 128, STAGE = SA + SM_ + 2 * SBH;
  constexpr int A_N = BM / 8, M_N = 1, B_N = 2 * (BN / 8), W = A_N + M_N + B_N;
  constexpr int SL = (W + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned wm = wave / WN, wn = wave % WN;
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = lid / tiles, trem = lid - b * tiles;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const size_t row_base = (size_t)b * p.sA;  // first blob row of this grid batch (sA counts rows here)
  const float* B = p.Bp ? p.Bp[b] : p.B + (size_t)b * p.sB;
  float* C = p.Cp ? p.Cp[b] : p.C + (size_t)b * p.sC;
  const int mlast = p.M - 1;

  const char* src[SL];
  size_t step[SL];
  unsigned loff[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    src[i] = nullptr; step[i] = 0; loff[i] = 0;
    if (t < (unsigned)A_N) {  // 8 blob rows x 128 B of kept values
      const unsigned row = 8u * t + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      src[i] = p.vals + (row_base + (size_t)gr) * 128 + 16u * cs;
      step[i] = p.Mtot * 128;
      loff[i] = t * 1024u;
    } else if (t == (unsigned)A_N) {  // metadata of the 128 rows: lane -> rows 2 lane, 2 lane + 1 (M even: whole pairs)
      int gr = m0 + 2 * (int)lane;
      gr = gr < mlast ? gr : (mlast & ~1);
      src[i] = p.meta + (row_base + (size_t)gr) * 8;
      step[i] = p.Mtot * 8;
      loff[i] = SA;
    } else if (t < (unsigned)W) {  // B: half h, 8 k-rows of one 32-column panel
      const unsigned j = t - A_N - M_N, h = j / (BN / 8), jj = j - h * (BN / 8), panel = jj >> 2;
      const unsigned kr = 8u * (jj & 3u) + (lane >> 3), cs = (lane & 7u) ^ (4u * ((kr >> 3) & 1u));
      int gc = n0 + (int)(32u * panel + 4u * cs);
      gc = gc <= p.N - 4 ? gc : p.N - 4;
      src[i] = reinterpret_cast<const char*>(B + (size_t)(32u * h + kr) * p.ldb + gc);
      step[i] = (size_t)64 * p.ldb * 4;
      loff[i] = SA + SM_ + h * SBH + panel * 4096u + (jj & 3u) * 1024u;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const unsigned t = wave + (unsigned)NW * i;  // wave-uniform
      if (t >= (unsigned)W) continue;
      __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)kt * step[i]), (lptr_t*)(base + loff[i]), 16, 0, 0);
    }
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  const int nkt = p.kc / 64;
  if (nkt > 0) stage(0, 0);
  const unsigned g = lane >> 4, r = lane & 15u;
  for (int kt = 0; kt < nkt; ++kt) {
    wait_dma_and_barrier<0>();  // ring of 2: nothing newer than this stage is in flight
    if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
    const char* As = smem + (kt & 1) * STAGE;
    const char* Ms = As + SA;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const char* Bs = Ms + SM_ + h * SBH;
      float ad[FM][8];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const unsigned row = wm * TM + i * 16 + r;
        const f4 kv = *reinterpret_cast<const f4*>(As + a_off(row, 4u * h + g));
        const unsigned mb = *reinterpret_cast<const unsigned char*>(Ms + row * 8u + 4u * h + g);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const unsigned nib = (mb >> (4 * st)) & 0xfu, p0 = nib & 3u, p1 = nib >> 2;
#pragma unroll
          for (unsigned t = 0; t < 4; ++t) ad[i][4 * st + t] = t == p0 ? kv[2 * st] : (t == p1 ? kv[2 * st + 1] : 0.0f);
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float bv[8];
        const unsigned col = wn * TN + j * 16 + r;
#pragma unroll
        for (int s = 0; s < 8; ++s) bv[s] = *reinterpret_cast<const float*>(Bs + b32_off(8u * g + s, col));
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int i = 0; i < FM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[s], ad[i][s], acc[i][j], 0, 0, 0);
      }
    }
  }

  const bool c_vec = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int gr = m0 + (int)(wm * TM + i * 16 + r);
      const int gc = n0 + (int)(wn * TN + j * 16 + 4u * g);
      if (gr >= p.M || gc >= p.N) continue;
      float* dst = C + (size_t)gr * p.ldc + gc;
      f4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = p.alpha * acc[i][j][q];
      if (c_vec && gc + 4 <= p.N) {
        if (p.beta != 0.0f) {
          const f4 old = *reinterpret_cast<const f4*>(dst);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += p.beta * old[q];
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (gc + q < p.N) dst[q] = p.beta != 0.0f ? v[q] + p.beta * dst[q] : v[q];
      }
    }
}

template <int BN, int WM, int WN>
static int launch_spmma32_dma(const Gemm32Args& a0, hipStream_t st) {
  Gemm32Args a = a0;
  a.tiles_m = (a.M + 127) / 128;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("spmma_f32: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds = 2 * ((size_t)128 * 128 + 128 * 8 + 2 * (size_t)BN * 128);
  static LdsOptIn lds_optin;
  if (lds > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_f32_dma_kernel<BN, WM, WN>), lds, "spmma_f32_dma_kernel")) return rc;
  }
  spmma_f32_dma_kernel<BN, WM, WN><<<dim3((unsigned)nwg), dim3(64 * WM * WN), lds, st>>>(a);
  return check_launch("spmma_f32_dma_kernel");
}


// Tile shape: the kernel is bound by the fp32 matrix pipe, not by operand traffic, so small tiles cost nothing per
// flop (4096^3: 104.7 / 111.1 / 101.7 / 105.3 TF/s for 128x128 / 128x64 / 64x128 / 64x64) and win whenever the
// tile count is not a large multiple of the CU count: 64 x 64 is the best shape on every ResNet-18 layer at
// b = 32 (5.34 ms for the table against 8.25 / 6.01 / 7.18 ms, tools/archive/f32_probe.py and tools/sweep.py
// --dtype f32).  128 x 64 takes over once a CU gets >= 32 of the small tiles.
template <int MODE>
static int dispatch32(const Gemm32Args& a, hipStream_t st) {
  struct Cand { int bm, bn; };
  static const Cand cands[4] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  static const int force = tuning_int("SM_GEMM32_CFG", -1);  // tuning aid: candidate index
  const double cus = (double)device_cu_count();
  const double small_tiles = (double)((a.M + 63) / 64) * (double)((a.N + 63) / 64) * a.batch;
  int best = small_tiles / cus >= 32.0 ? 1 : 3;
  if (force >= 0 && force < 4) best = force;
  if constexpr (MODE == 0 || MODE == 2) {
    // LDS-DMA pipeline when whole 16-byte chunks can be moved: K % 32 == 0, aligned rows (pointer-array batches: the
    // caller's bases, hipMalloc gives 256 B); SM_GEMM32_DMA=0 (tuning aid) keeps the register-staged kernel
    static const int dma_env = tuning_int("SM_GEMM32_DMA", 1);
    const bool a_ok = (a.lda % 4 == 0) && (a.sA % 4 == 0) && (a.Ap || (reinterpret_cast<uintptr_t>(a.A) & 15u) == 0);
    const bool b_ok = (a.ldb % 4 == 0) && (a.sB % 4 == 0) && (a.Bp || (reinterpret_cast<uintptr_t>(a.B) & 15u) == 0);
    if (dma_env && a.K % 32 == 0 && a.K >= 32 && a_ok && b_ok && (MODE == 2 || (a.N % 4 == 0 && a.N >= 4))) {
      constexpr bool BKM = MODE == 2;
      const int cfg = dma_env;  // 1: default choice; 2..: forced shapes for tuning
      if (cfg == 2) return launch32_dma<64, 64, 2, 2, 2, BKM>(a, st);
      if (cfg == 3) return launch32_dma<64, 64, 2, 2, 3, BKM>(a, st);
      if (cfg == 4) return launch32_dma<128, 64, 4, 1, 2, BKM>(a, st);
      if (cfg == 5) return launch32_dma<128, 128, 2, 2, 2, BKM>(a, st);
      if (cfg == 6) return launch32_dma<128, 128, 2, 4, 2, BKM>(a, st);
      if (cfg == 7) return launch32_dma<128, 64, 4, 1, 3, BKM>(a, st);
      if (cfg == 8) return launch32_dma<128, 128, 2, 4, 3, BKM>(a, st);
      if (cfg == 9) return launch32_dma<128, 128, 4, 4, 2, BKM>(a, st);
      // short operands get tiles as short as they are; narrow outputs 128 x 64 tiles (more of them); otherwise
      // 128 x 128 over 16 waves (measured on the ResNet
      // shapes and on 4096^3 / 8192^2 x 2048: tools/archive/f32_probe.py under SM_GEMM32_DMA=2..9)
      if (cfg == 10) return launch32_dma<64, 128, 1, 4, 2, BKM>(a, st);
      if (a.M <= 64) return a.N <= 64 ? launch32_dma<64, 64, 2, 2, 2, BKM>(a, st) : launch32_dma<64, 128, 1, 4, 2, BKM>(a, st);
      return a.N <= 128 ? launch32_dma<128, 64, 4, 1, 2, BKM>(a, st) : launch32_dma<128, 128, 4, 4, 2, BKM>(a, st);
    }
  }
  static const bool verbose = tuning_env("SM_GEMM32_VERBOSE") != nullptr;  // tuning aid
  if (verbose) fprintf(stderr, "gemm_f32 %d x %d x %d b=%d on %d CUs -> tile %d x %d\n", a.M, a.N, a.K, a.batch, (int)cus, cands[best].bm, cands[best].bn);
  switch (best) {
    case 1: return launch32<128, 64, 4, 1, MODE>(a, st);
    case 2: return launch32<64, 128, 1, 4, MODE>(a, st);
    case 3: return launch32<64, 64, 2, 2, MODE>(a, st);
    default: return launch32<128, 128, 2, 2, MODE>(a, st);
  }
}

// C^T[n x m] (row-major, ldc = m: i.e. column-major m x n C) = alpha * Bt[n x k] * Adense[m x k]^T + beta * C,
// Bt = the column-major k x n B of the reference's SpMM (row-major n x k), Adense row-major m x k.
// With Cptrs (a DEVICE array of `batch` C pointers) the grid covers all batches: Adense then holds `batch`
// consecutive m x k matrices and Bcm is shared.
int gemm_f32_colmajor_c_from_rowmajor_a(const float* Adense, const float* Bcm, float* Ccm, float* const* Cptrs,
                                        size_t m, size_t n, size_t k, size_t batch, float alpha, float beta,
                                        hipStream_t st) {
  Gemm32Args a = {};
  a.A = Bcm; a.B = Adense; a.C = Ccm; a.Cp = Cptrs;
  a.sA = 0; a.sB = m * k; a.sC = 0;
  a.M = (int)n; a.N = (int)m; a.K = (int)k;
  a.lda = (int)k; a.ldb = (int)k; a.ldc = (int)m;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  return dispatch32<2>(a, st);
}

// ---- fp64: plain FMA tiles, 64 x 64 outputs per 256-thread workgroup, 4 x 4 per thread ------------
struct Gemm64Args {
  const double* const* Ap;
  const double* const* Bp;
  double* const* Cp;
  int M, N, K, lda, ldb, ldc;  // row-major view (see sm_gemm_batched_f64)
  int ta, tb;                  // 1: A given M-contiguous / B given K-contiguous (the transposed operands)
  double alpha, beta;
};
__global__ __launch_bounds__(256) void gemm_f64_kernel(const Gemm64Args p) {
  __shared__ double As[64][17], Bs[16][65];
  const unsigned tid = threadIdx.x, tx = tid & 15u, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const double* A = p.Ap[blockIdx.z];
  const double* B = p.Bp[blockIdx.z];
  double* C = p.Cp[blockIdx.z];
  double acc[4][4] = {};
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    for (unsigned q = tid; q < 64 * 16; q += 256) {
      const int r = q >> 4, c = q & 15;
      As[r][c] = (m0 + r < p.M && k0 + c < p.K)
                     ? (p.ta ? A[(size_t)(k0 + c) * p.lda + m0 + r] : A[(size_t)(m0 + r) * p.lda + k0 + c]) : 0.0;
    }
    for (unsigned q = tid; q < 16 * 64; q += 256) {
      const int r = q >> 6, c = q & 63;
      Bs[r][c] = (k0 + r < p.K && n0 + c < p.N)
                     ? (p.tb ? B[(size_t)(n0 + c) * p.ldb + k0 + r] : B[(size_t)(k0 + r) * p.ldb + n0 + c]) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      double a[4], bb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[ty * 4 + i][kk];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], bb[j], acc[i][j]);
    }
    __syncthreads();
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const int r = m0 + ty * 4 + i, c = n0 + tx * 4 + j;
      if (r < p.M && c < p.N) {
        double* d = C + (size_t)r * p.ldc + c;
        *d = p.beta != 0.0 ? p.alpha * acc[i][j] + p.beta * *d : p.alpha * acc[i][j];
      }
    }
}

// ---- fp64 on the matrix cores: v_mfma_f64_16x16x4_f64 on the fp32 kernel's LDS-DMA pipeline (K % 16 == 0, even N).
// Stage = 16 k = 128 B per A row; B in panels of 16 columns [16 k][128 B] (chunk c of k-row kr at c ^ 4 * ((kr >> 2) & 1));
// the instruction's four k-slots carry k = 4 g + s at step s, so a lane owns 4 consecutive k of its A row per stage (two
// ds_read_b128).  Operands swapped as in the fp32 kernels; the f64 result map then leaves lane l, register q with
// C[row l & 15][column (l >> 4) + 4 q] (cdna_hip_programming.md, "v_mfma_f64_16x16x4_f64").
typedef double d4v __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned b64_off(unsigned kr, unsigned col) {  // byte offset in the [panel][16][128 B] image
  return (col >> 4) * 2048u + kr * 128u + 16u * (((col & 15u) >> 1) ^ (4u * ((kr >> 2) & 1u))) + 8u * (col & 1u);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f64_dma_kernel(const Gemm64Args p) {
  constexpr int NW = WM * WN, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int SA = BM * 128, SB = BN * 128, STAGE = SA + SB;
  constexpr int A_N = BM / 8, B_N = BN / 8, W = A_N + B_N;  // 1 KiB DMA wave-instructions per stage
  static_assert(W % NW == 0, "equal DMA share per wave");
  constexpr int SL = W / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned wm = wave / WN, wn = wave % WN;
  const int m0 = (int)blockIdx.y * BM, n0 = (int)blockIdx.x * BN;
  const double* A = p.Ap[blockIdx.z];
  const double* B = p.Bp[blockIdx.z];
  double* C = p.Cp[blockIdx.z];
  const int mlast = p.M - 1;

  const char* src[SL];
  size_t step[SL];
  unsigned loff[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    if (t < (unsigned)A_N) {  // 8 rows x 128 B
      const unsigned row = 8u * t + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      src[i] = reinterpret_cast<const char*>(A + (size_t)gr * p.lda + 2u * cs);
      step[i] = 128;
      loff[i] = t * 1024u;
    } else {  // 8 k-rows of one 16-column panel
      const unsigned j = t - A_N, panel = j >> 1, kr = 8u * (j & 1u) + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ (4u * ((kr >> 2) & 1u));
      int gc = n0 + (int)(16u * panel + 2u * cs);
      gc = gc <= p.N - 2 ? gc : p.N - 2;
      src[i] = reinterpret_cast<const char*>(B + (size_t)kr * p.ldb + gc);
      step[i] = (size_t)16 * p.ldb * 8;
      loff[i] = SA + panel * 2048u + (j & 1u) * 1024u;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < SL; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)kt * step[i]), (lptr_t*)(base + loff[i]), 16, 0, 0);
  };

  d4v acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = d4v{0.0, 0.0, 0.0, 0.0};

  const int nkt = p.K / 16;
  if (nkt > 0) stage(0, 0);
  const unsigned g = lane >> 4, r = lane & 15u;
  for (int kt = 0; kt < nkt; ++kt) {
    wait_dma_and_barrier<0>();  // ring of 2
    if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
    const char* As = smem + (kt & 1) * STAGE;
    const char* Bs = As + SA;
    d2v alo[FM], ahi[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const unsigned row = wm * TM + i * 16 + r;
      alo[i] = *reinterpret_cast<const d2v*>(As + a_off(row, 2u * g));
      ahi[i] = *reinterpret_cast<const d2v*>(As + a_off(row, 2u * g + 1u));
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      double bv[4];
      const unsigned col = wn * TN + j * 16 + r;
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = *reinterpret_cast<const double*>(Bs + b64_off(4u * g + s, col));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < FM; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[s], s < 2 ? alo[i][s] : ahi[i][s - 2], acc[i][j], 0, 0, 0);
    }
  }

#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int gr = m0 + (int)(wm * TM + i * 16 + r);
      if (gr >= p.M) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gc = n0 + (int)(wn * TN + j * 16 + 4u * q + g);
        if (gc >= p.N) continue;
        double* d = C + (size_t)gr * p.ldc + gc;
        *d = p.beta != 0.0 ? p.alpha * acc[i][j][q] + p.beta * *d : p.alpha * acc[i][j][q];
      }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch64_dma(const Gemm64Args& a, size_t batch, hipStream_t st) {
  constexpr size_t lds = 2 * (size_t)(BM + BN) * 128;
  static LdsOptIn lds_optin;
  if (lds > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&gemm_f64_dma_kernel<BM, BN, WM, WN>), lds, "gemm_f64_dma_kernel")) return rc;
  }
  dim3 grid((unsigned)ceil_div(a.N, BN), (unsigned)ceil_div(a.M, BM), (unsigned)batch);
  gemm_f64_dma_kernel<BM, BN, WM, WN><<<grid, dim3(64 * WM * WN), lds, st>>>(a);
  return check_launch("gemm_f64_dma_kernel");
}

}  // namespace sm

using namespace sm;

extern "C" {

int sm_gemm_rowmajor_f32(const float* A, const float* B, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                         size_t strideA, size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream) {
  if (!A || !B || !C || lda < k) {
    set_error("sm_gemm_rowmajor_f32: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || lda > 0x7fffffffull) {
    set_error("sm_gemm_rowmajor_f32: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  Gemm32Args a = {};
  a.A = A; a.B = B; a.C = C;
  a.M = (int)m; a.N = (int)n; a.K = (int)k;
  a.lda = (int)lda; a.ldb = (int)n; a.ldc = (int)n;
  a.sA = strideA; a.sB = strideB; a.sC = strideC;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  if (batch > 1 && strideB == 0 && strideA == m * lda && strideC == m * n) {
    a.M = (int)(m * batch);
    a.batch = 1;
  }
  return dispatch32<0>(a, (hipStream_t)stream);
}

int sm_spmma_fused_f32(const float* A, const float* B, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                       size_t strideA, size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream) {
  if (!A || !B || !C || lda < k) {
    set_error("sm_spmma_fused_f32: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || lda > 0x7fffffffull) {
    set_error("sm_spmma_fused_f32: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  // whole 32-deep stages of 16-byte aligned rows only (the LDS-DMA pipeline); anything else: sm_compress24_f32 + sm_spmma_f32
  if (k % 32 != 0 || n % 4 != 0 || lda % 4 != 0 || strideA % 4 != 0 || strideB % 4 != 0 || !aligned16(A) || !aligned16(B)) {
    set_error("sm_spmma_fused_f32: needs k %% 32 == 0, n %% 4 == 0 and 16-byte aligned rows (use the staged path)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  Gemm32Args a = {};
  a.A = A; a.B = B; a.C = C;
  a.M = (int)m; a.N = (int)n; a.K = (int)k;
  a.lda = (int)lda; a.ldb = (int)n; a.ldc = (int)n;
  a.sA = strideA; a.sB = strideB; a.sC = strideC;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  if (batch > 1 && strideB == 0 && strideA == m * lda && strideC == m * n) {
    a.M = (int)(m * batch);
    a.batch = 1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (a.M <= 64) return a.N <= 64 ? launch32_dma<64, 64, 2, 2, 2, false, 1>(a, st) : launch32_dma<64, 128, 1, 4, 2, false, 1>(a, st);
  // n <= 128: 128 x 64 tiles with one wave column -> the register form (P24 = 2: every strip selected once, by the lane that
  // feeds it, no LDS pass, no extra barrier); wider: 16 waves in a 4 x 4 grid, where the register form would repeat the selection
  // per wave column -> the in-LDS form.  Round 4, ResNet-18 table: 4.47 -> 4.37 ms (0.85 -> 0.87 x the dense fp32 GEMM);
  // one-wave-column tilings for n > 128 (128 x 128 x 4 or 8 waves) measured no better (4.43-4.48 ms).
  return a.N <= 128 ? launch32_dma<128, 64, 4, 1, 2, false, 2>(a, st) : launch32_dma<128, 128, 4, 4, 2, false, 1>(a, st);
}

int sm_gemm_batched_f32(const float* const* A_ptrs, const float* const* B_ptrs, float* const* C_ptrs, size_t m, size_t n,
                        size_t k, size_t batch, int ta, int tb, float alpha, float beta, sm_stream_t stream) {
  if (!A_ptrs || !B_ptrs || !C_ptrs) {
    set_error("sm_gemm_batched_f32: null pointer array");
    return SM_STATUS_INVALID_VALUE;
  }
  if ((ta != SM_OP_N && ta != SM_OP_T) || (tb != SM_OP_N && tb != SM_OP_T)) {
    set_error("sm_gemm_batched_f32: operation must be SM_OP_N or SM_OP_T");
    return SM_STATUS_INVALID_VALUE;
  }
  if ((ta == SM_OP_T && m < k) || (tb == SM_OP_T && k < n)) {  // see sm_gemm_batched_f16
    set_error("sm_gemm_batched_f32: leading dimension (lda = m, ldb = k) shorter than a column of the transposed operand");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || batch > 0x7fffffffull) {
    set_error("sm_gemm_batched_f32: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  // column-major C = A*B  <=>  row-major C^T[n x m] = B^T[n x k] * A^T[k x m]
  Gemm32Args a = {};
  a.Ap = B_ptrs; a.Bp = A_ptrs; a.Cp = C_ptrs;
  a.M = (int)n; a.N = (int)m; a.K = (int)k;
  a.lda = (int)k; a.ldb = (int)m; a.ldc = (int)m;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  // op(B) = T: the row-major view's A operand arrives n-contiguous (ATR); op(A) = T: its B operand k-contiguous (MODE 2)
  if (tb == SM_OP_T) return ta == SM_OP_T ? launch32<64, 64, 2, 2, 2, true>(a, (hipStream_t)stream)
                                          : launch32<64, 64, 2, 2, 0, true>(a, (hipStream_t)stream);
  if (ta == SM_OP_T) return dispatch32<2>(a, (hipStream_t)stream);
  return dispatch32<0>(a, (hipStream_t)stream);
}

int sm_spmma_f32(const void* blob, const float* B, float* C, size_t m, size_t n, size_t k, size_t batch, size_t strideB,
                 size_t strideC, float alpha, float beta, sm_stream_t stream) {
  if (!blob || !B || !C) {
    set_error("sm_spmma_f32: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull) {
    set_error("sm_spmma_f32: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  const BlobLayout L = blob_layout(m, k, 4, batch);
  Gemm32Args a = {};
  a.vals = (const char*)blob;
  a.meta = (const char*)blob + L.meta_off;
  a.Mtot = L.M;
  a.B = B; a.C = C;
  a.M = (int)m; a.N = (int)n; a.K = (int)k; a.kc = (int)L.kc;
  a.ldb = (int)n; a.ldc = (int)n;
  a.sA = m; a.sB = strideB; a.sC = strideC;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  if (batch > 1 && strideB == 0 && strideC == m * n) {
    a.M = (int)(m * batch);
    a.batch = 1;
  }
  // LDS-DMA pipeline: whole 64-k planes, row pairs (metadata moves as 16-byte pairs), whole 16-byte B chunks
  static const int dma_env = tuning_int("SM_SPMMA32_DMA", 1);  // tuning aid: 0 = register-staged kernel
  const bool pairs = (a.batch == 1 ? (a.M % 2 == 0) : (m % 2 == 0));
  if (dma_env && k % 64 == 0 && k >= 64 && pairs && n % 4 == 0 && n >= 4 && strideB % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(B) & 15u) == 0) {
    if (dma_env == 3) return launch_spmma32_dma<128, 2, 4>(a, (hipStream_t)stream);
    if (dma_env == 5) return launch_spmma32_dma<64, 4, 2>(a, (hipStream_t)stream);
    // 128 x 64 tiles on every shape (two workgroups per CU; the 128 x 128 forms hold one and measured 25-40 % slower)
    return launch_spmma32_dma<64, 4, 1>(a, (hipStream_t)stream);
  }
  return dispatch32<1>(a, (hipStream_t)stream);
}

int sm_gemm_batched_f64(const double* const* A_ptrs, const double* const* B_ptrs, double* const* C_ptrs, size_t m, size_t n,
                        size_t k, size_t batch, int ta, int tb, double alpha, double beta, sm_stream_t stream) {
  if (!A_ptrs || !B_ptrs || !C_ptrs) {
    set_error("sm_gemm_batched_f64: null pointer array");
    return SM_STATUS_INVALID_VALUE;
  }
  if ((ta != SM_OP_N && ta != SM_OP_T) || (tb != SM_OP_N && tb != SM_OP_T)) {
    set_error("sm_gemm_batched_f64: operation must be SM_OP_N or SM_OP_T");
    return SM_STATUS_INVALID_VALUE;
  }
  if ((ta == SM_OP_T && m < k) || (tb == SM_OP_T && k < n)) {  // see sm_gemm_batched_f16
    set_error("sm_gemm_batched_f64: leading dimension (lda = m, ldb = k) shorter than a column of the transposed operand");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || batch > 65535) {
    set_error("sm_gemm_batched_f64: dimension not supported");
    return SM_STATUS_NOT_SUPPORTED;
  }
  Gemm64Args a = {};
  a.Ap = B_ptrs; a.Bp = A_ptrs; a.Cp = C_ptrs;  // transposed view, as for fp32
  a.M = (int)n; a.N = (int)m; a.K = (int)k;
  a.lda = (int)k; a.ldb = (int)m; a.ldc = (int)m;
  a.ta = tb == SM_OP_T; a.tb = ta == SM_OP_T;  // roles swap with the operands
  a.alpha = alpha; a.beta = beta;
  // matrix-core path: whole 16-k stages of 16-byte chunks (even leading dimensions; the pointer arrays live on the
  // device, their bases are the caller's -- a 16-byte global access needs dword alignment only); SM_GEMM64_DMA=0
  // (tuning aid) keeps the FMA kernel
  static const int dma_env = tuning_int("SM_GEMM64_DMA", 1);
  if (dma_env && !a.ta && !a.tb && a.K % 16 == 0 && a.K >= 16 && a.N % 2 == 0 && a.N >= 2 && a.lda % 2 == 0 && a.ldb % 2 == 0 &&
      ceil_div((size_t)a.M, (size_t)64) <= 65535) {
    if (dma_env == 2) return launch64_dma<64, 64, 2, 2>(a, batch, (hipStream_t)stream);
    if (dma_env == 5) return launch64_dma<128, 128, 4, 4>(a, batch, (hipStream_t)stream);
    // 128 x 128 tiles over 16 waves once they fill 3/4 of the CUs, 64 x 64 otherwise (tools/archive/f64_probe.py under SM_GEMM64_DMA=2..5)
    const size_t big_tiles = ceil_div((size_t)a.M, (size_t)128) * ceil_div((size_t)a.N, (size_t)128) * batch;
    if (a.M > 64 && a.N > 64 && 4 * big_tiles >= 3 * (size_t)device_cu_count())
      return launch64_dma<128, 128, 4, 4>(a, batch, (hipStream_t)stream);
    return launch64_dma<64, 64, 2, 2>(a, batch, (hipStream_t)stream);
  }
  dim3 grid((unsigned)ceil_div(a.N, 64), (unsigned)ceil_div(a.M, 64), (unsigned)batch);
  gemm_f64_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>(a);
  return check_launch("gemm_f64_kernel");
}

}  // extern "C"
