// gemm_f16.hip -- dense fp16 GEMM on v_mfma_f32_16x16x32_f16 (fp32 accumulate, one rounding).
// It is the metric's denominator: the hand-written replacement of cublasHgemmBatched
// (include/sparsify.me/gemm.hxx:80-81) and the like-for-like dense form of the 2:4 kernel.
//
// One kernel computes the row-major product  C[M x N] = alpha * A[M x K] * B[K x N] + beta * C
// (A rows K-contiguous, B rows N-contiguous).  The reference's column-major batched GEMM
// (lda = m, ldb = k, ldc = m) is the same memory seen transposed: C^T[n x m] = B^T[n x k] A^T[k x m],
// all three row-major, so sm_gemm_batched_f16 launches this kernel with (M, N, K) = (n, m, k) and the
// operand roles swapped -- no data movement.
//
// Structure: 256 threads = 4 waves, BM x BN output tile, BK = 64.  Global -> registers (16 B per
// lane, zero-filled out of range) -> swizzled LDS images (mma_tile.h); the next K-tile's global
// loads are issued before the current tile's MFMAs.  The MFMA is issued with its operands swapped
// (B fragment as srcA) so each lane ends up with four CONSECUTIVE n of one row; the epilogue packs
// them to 8 bytes, stages the tile in LDS and writes C with 16-byte row-contiguous stores.
#include "mma_tile.h"
#include "spmma_args.h"

namespace sm {

struct GemmArgs {
  const half_t* A;
  const half_t* B;
  half_t* C;
  const half_t* const* Ap;  // optional device pointer arrays (batch entries); null -> base + b*stride
  const half_t* const* Bp;
  half_t* const* Cp;
  size_t sA, sB, sC;        // batch strides in elements
  int M, N, K;
  int lda, ldb, ldc;
  int batch, tiles_m, tiles_n;
  float alpha, beta;
  // F32OUT instantiation only (sm::gemm_f16_f32out, the dense-MFMA form of the COO SpMM): fp32 C, and an A operand whose
  // k runs over [0, a_wrap_kt * 64) TWICE (stage kt >= a_wrap_kt re-reads stage kt - a_wrap_kt) against a B of 2 x that depth
  float* C32;
  int a_wrap_kt;
  // (round 4) device-side scalars of the COO dense-MFMA form: the result is multiplied by alpha_dev[0] and alpha_dev[1] (the
  // inverses of the operands' power-of-two scales, computed on the device), and a workgroup returns at once when *skip_flag != 0 (an operand left the
  // fp16 range: C must stay untouched for the exact kernels that run instead)
  const float* alpha_dev;
  const int* skip_flag;
};

// 8 halves starting at p[col]; elements at or beyond `limit` columns read as zero.
// VEC = guaranteed alignment of rowp + col in halves: 8 (one 16-byte load), 4 (8-byte loads), 1.
template <int VEC>
__device__ __forceinline__ u4 load_chunk(const half_t* rowp, int col, int limit, bool row_ok) {
  u4 v = {0u, 0u, 0u, 0u};
  if (!row_ok || col >= limit) return v;
  if (VEC == 8 && col + 8 <= limit) return *reinterpret_cast<const u4*>(rowp + col);
  if (VEC == 4 && col + 4 <= limit) {
    const u2 lo = *reinterpret_cast<const u2*>(rowp + col);
    u2 hi = {0u, 0u};
    if (col + 8 <= limit) {
      hi = *reinterpret_cast<const u2*>(rowp + col + 4);
    } else if (col + 4 < limit) {
      h4 e;
#pragma unroll
      for (int t = 0; t < 4; ++t) e[t] = (col + 4 + t < limit) ? rowp[col + 4 + t] : (half_t)0.0f;
      hi = __builtin_bit_cast(u2, e);
    }
    return u4{lo[0], lo[1], hi[0], hi[1]};
  }
  h8 e;
#pragma unroll
  for (int t = 0; t < 8; ++t) e[t] = (col + t < limit) ? rowp[col + t] : (half_t)0.0f;
  return __builtin_bit_cast(u4, e);
}

// TA: the A operand is given M-contiguous (element (r, kk) at A[kk * lda + r]); TB: the B operand is given K-contiguous
// (element (kk, c) at B[c * ldb + kk]) -- the transposed operands of sm_gemm_batched_f16.  The 16-byte chunks then run
// along the contiguous direction and are scattered into the same LDS images with 2-byte writes.
template <int BM, int BN, int WM, int WN, int VEC, bool TA = false, bool TB = false, bool BF = false>
__global__ __launch_bounds__(256) void gemm_f16_kernel(const GemmArgs p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int A_CH = BM * 8 / 256;   // 16-byte chunks of the A tile per thread
  constexpr int B_CH = BN / 32;        // 64 k-rows x BN/8 chunks / 256 threads
  constexpr int CPITCH = BN * 2 + 16;  // bytes per row of the epilogue image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Bs = smem + BM * 128;

  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned wm = wave / WN, wn = wave % WN;

  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = lid / tiles, trem = lid - b * tiles;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;

  const half_t* A = p.Ap ? p.Ap[b] : p.A + (size_t)b * p.sA;
  const half_t* B = p.Bp ? p.Bp[b] : p.B + (size_t)b * p.sB;
  half_t* C = p.Cp ? p.Cp[b] : p.C + (size_t)b * p.sC;

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  u4 ra[A_CH], rb[B_CH];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      if constexpr (TA) {
        const unsigned q = tid + 256u * i, kk = q / (BM / 8), rc = q % (BM / 8);
        const int gk = k0 + (int)kk;
        ra[i] = load_chunk<VEC>(A + (size_t)gk * p.lda, m0 + 8 * (int)rc, p.M, gk < p.K);
      } else {
        const unsigned q = tid + 256u * i, row = q >> 3, ch = q & 7u;
        const int gr = m0 + (int)row;
        ra[i] = load_chunk<VEC>(A + (size_t)gr * p.lda, k0 + 8 * (int)ch, p.K, gr < p.M);
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      if constexpr (TB) {
        const unsigned q = tid + 256u * i, col = q >> 3, kc = q & 7u;
        const int gn = n0 + (int)col;
        rb[i] = load_chunk<VEC>(B + (size_t)gn * p.ldb, k0 + 8 * (int)kc, p.K, gn < p.N);
      } else {
        const unsigned q = tid + 256u * i, kr = q / (BN / 8), cn = q % (BN / 8);
        const int gk = k0 + (int)kr;
        rb[i] = load_chunk<VEC>(B + (size_t)gk * p.ldb, n0 + 8 * (int)cn, p.N, gk < p.K);
      }
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      if constexpr (TA) {
        const unsigned q = tid + 256u * i, kk = q / (BM / 8), rc = q % (BM / 8);
        const h8 e = __builtin_bit_cast(h8, ra[i]);
#pragma unroll
        for (unsigned t = 0; t < 8; ++t) *reinterpret_cast<half_t*>(As + a_off(8u * rc + t, kk >> 3) + 2u * (kk & 7u)) = e[t];
      } else {
        const unsigned q = tid + 256u * i, row = q >> 3, ch = q & 7u;
        *reinterpret_cast<u4*>(As + a_off(row, ch)) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      if constexpr (TB) {
        const unsigned q = tid + 256u * i, col = q >> 3, kc = q & 7u;
        const h8 e = __builtin_bit_cast(h8, rb[i]);
#pragma unroll
        for (unsigned t = 0; t < 8; ++t) *reinterpret_cast<half_t*>(Bs + b_off<64>(8u * kc + t, col)) = e[t];
      } else {
        const unsigned q = tid + 256u * i, kr = q / (BN / 8), cn = q % (BN / 8);
        *reinterpret_cast<u4*>(Bs + b_off<64>(kr, 8u * cn)) = rb[i];
      }
    }
  };

  const int nkt = (p.K + 63) / 64;
  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < nkt) gload((kt + 1) * 64);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h8 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const unsigned row = wm * TM + i * 16 + (lane & 15u);
        af[i] = *reinterpret_cast<const h8*>(As + a_off(row, 4u * s + (lane >> 4)));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const unsigned col0 = wn * TN + j * 16, kr0 = 32u * s + 8u * (lane >> 4);
        const s4 lo = b_read_tr<64>(Bs, kr0, col0, lane);
        const s4 hi = b_read_tr<64>(Bs, kr0 + 4u, col0, lane);
        typedef short s8 __attribute__((ext_vector_type(8)));
        const s8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bf[j] = __builtin_bit_cast(h8, both);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // operands swapped: D = B^T-fragment x A^T-fragment = (A*B)^T tile, i.e. this lane holds
          // C[row (lane&15)][cols 4*(lane>>4) .. +3] of fragment (i, j)
          acc[i][j] = mfma16<BF>(bf[j], af[i], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue
  const bool c_vec = VEC == 8 && (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
  if (p.beta == 0.0f && c_vec) {
    char* Cs = smem;  // [BM][CPITCH]
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const unsigned row = wm * TM + i * 16 + (lane & 15u), col = wn * TN + j * 16 + 4u * (lane >> 4);
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = to_elt<BF>(p.alpha * acc[i][j][r]);
        *reinterpret_cast<h4*>(Cs + row * CPITCH + col * 2) = o;
      }
    __syncthreads();
    constexpr int C_CH = BM * (BN / 8) / 256;
#pragma unroll
    for (int i = 0; i < C_CH; ++i) {
      const unsigned q = tid + 256u * i, row = q / (BN / 8), cn = q % (BN / 8);
      const int gr = m0 + (int)row, gc = n0 + 8 * (int)cn;
      if (gr >= p.M || gc >= p.N) continue;
      const u4 v = *reinterpret_cast<const u4*>(Cs + row * CPITCH + cn * 16);
      half_t* dst = C + (size_t)gr * p.ldc + gc;
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
    // general alpha/beta or unaligned C: straight from the accumulators, one rounding
    const bool c8 = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 7u) == 0);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int gr = m0 + (int)(wm * TM + i * 16 + (lane & 15u));
        const int gc = n0 + (int)(wn * TN + j * 16 + 4u * (lane >> 4));
        if (gr >= p.M) continue;
        if (p.beta == 0.0f && c8 && gc + 4 <= p.N) {  // four consecutive n of one row: one 8-byte store
          h4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = to_elt<BF>(p.alpha * acc[i][j][r]);
          *reinterpret_cast<h4*>(C + (size_t)gr * p.ldc + gc) = o;
          continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (gc + r >= p.N) continue;
          half_t* dst = C + (size_t)gr * p.ldc + gc + r;
          float v = p.alpha * acc[i][j][r];
          if (p.beta != 0.0f) v += p.beta * to_f32<BF>(*dst);
          *dst = to_elt<BF>(v);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Fast path (K % 64 == 0, N % 8 == 0, rows of A and B 8-byte aligned or better): the same LDS-DMA
// ring as the 2:4 kernel (spmma_f16.hip).  Stage = 64 k: A [BM][128 B] + B [64][BN]; images are
// lane-linear for the DMA with the swizzle on the per-lane source address.  Edge rows / columns are
// clamped to the last valid one (their products land in outputs that are never stored).
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NS, bool BF = false, bool F32OUT = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f16_dma_kernel(const GemmArgs p) {
  constexpr int NW = WM * WN;
  static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(FM >= 1 && FN >= 1, "wave tile");
  constexpr int SA = BM * 128, SB = 64 * BN * 2, STAGE = SA + SB;
  constexpr int A_N = BM / 8, B_N = BN / 8, W = A_N + B_N;  // 1 KiB DMA wave-instructions per stage
  constexpr int SL = (W + NW - 1) / NW;
  constexpr int LPS = W / NW;
  static_assert(LPS >= 1, "every wave must issue at least one DMA per stage");
  constexpr int CPITCH = BN * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (F32OUT) {  // (a scalar load: the same for every wave of the grid, before any DMA is issued)
    if (p.skip_flag && *p.skip_flag != 0) return;
  }

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned wm = wave / WN, wn = wave % WN;
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = lid / tiles, trem = lid - b * tiles;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;

  const half_t* A = p.Ap ? p.Ap[b] : p.A + (size_t)b * p.sA;
  const half_t* B = p.Bp ? p.Bp[b] : p.B + (size_t)b * p.sB;
  half_t* C = p.Cp ? p.Cp[b] : p.C + (size_t)b * p.sC;
  const int mlast = p.M - 1;

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
      src[i] = reinterpret_cast<const char*>(A + (size_t)gr * p.lda + 8u * cs);
      step[i] = 128;
      loff[i] = t * 1024u;
    } else {
      const unsigned j = t - A_N, panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
      const unsigned cs = (lane & 7u) ^ b_swz(kr);
      int gc = n0 + (int)(64u * panel + 8u * cs);
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      src[i] = reinterpret_cast<const char*>(B + (size_t)kr * p.ldb + gc);
      step[i] = (size_t)64 * p.ldb * 2;
      loff[i] = SA + panel * 8192u + (j & 7u) * 1024u;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
    int kta = kt;
    if constexpr (F32OUT) kta = (p.a_wrap_kt > 0 && kt >= p.a_wrap_kt) ? kt - p.a_wrap_kt : kt;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const unsigned t = wave + (unsigned)NW * i;
      if (t >= (unsigned)W) continue;
      const int ks = (F32OUT && t < (unsigned)A_N) ? kta : kt;
      __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + (size_t)ks * step[i]), (lptr_t*)(base + loff[i]), 16, 0, 0);
    }
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / 64;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) stage(s, s);
  const unsigned g = lane >> 4, r = lane & 15u;
  int cur = 0, fill = NS - 1;
  for (int kt = 0; kt < nkt; ++kt) {
    const int ahead = (nkt - 1 - kt) < (NS - 2) ? (nkt - 1 - kt) : (NS - 2);
    if (NS >= 4 && ahead == 2) wait_dma_and_barrier<2 * LPS>();
    else if (NS >= 3 && ahead == 1) wait_dma_and_barrier<LPS>();
    else wait_dma_and_barrier<0>();
    if (kt + NS - 1 < nkt) stage(kt + NS - 1, fill);
    const char* As = smem + cur * STAGE;
    const char* Bs = As + SA;
    const unsigned bs_addr = (unsigned)(uintptr_t)(lds_char*)Bs;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h8 af[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const unsigned row = wm * TM + i * 16 + r;
        af[i] = *reinterpret_cast<const h8*>(As + a_off(row, 4u * s + g));
      }
      s4 lo[2], hi[2];
      auto issue = [&](int j, s4& v0, s4& v1) {
        const unsigned col0 = wn * TN + j * 16, q = r >> 2, pp = r & 3u;
        const unsigned a = bs_addr + b_off<64>(32u * s + 8u * g + q, col0 + 4u * pp);
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512"
                     : "=&v"(v0), "=&v"(v1) : "v"(a) : "memory");
      };
      issue(0, lo[0], hi[0]);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int c = j & 1, n = c ^ 1;
        if (j + 1 < FN) {
          issue(j + 1, lo[n], hi[n]);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo[c]), "+v"(hi[c]) :: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[c]), "+v"(hi[c]) :: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        typedef short s8 __attribute__((ext_vector_type(8)));
        const s8 both = {lo[c][0], lo[c][1], lo[c][2], lo[c][3], hi[c][0], hi[c][1], hi[c][2], hi[c][3]};
        const h8 bf = __builtin_bit_cast(h8, both);
#pragma unroll
        for (int i = 0; i < FM; ++i)
          acc[i][j] = mfma16<BF>(bf, af[i], acc[i][j]);  // swapped: C^T layout
      }
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
    fill = fill + 1 == NS ? 0 : fill + 1;
  }
  __syncthreads();

  if constexpr (F32OUT) {
    // fp32 output straight from the accumulators: a lane holds 4 consecutive columns of one row = one 16-byte store
    // (ldc % 4 == 0 and N % 4 == 0 are the launcher's conditions, so a 4-column piece is all in or all out)
    float* C32 = p.C32 + (size_t)b * p.sC;
    // two exact power-of-two factors applied one after the other: their product alone need not be a normal float
    const float alpha_ = p.alpha_dev ? p.alpha * p.alpha_dev[0] : p.alpha, post_ = p.alpha_dev ? p.alpha_dev[1] : 1.0f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int gr = m0 + (int)(wm * TM + i * 16 + r);
        int gc = n0 + (int)(wn * TN + j * 16 + 4u * g);
        if (gr >= p.M) continue;
        const int chunk0 = gc & ~7;  // N % 8 == 4: the clamped last chunk holds the outputs of columns N-8 .. N-1 (see below)
        if (chunk0 < p.N && chunk0 > p.N - 8) {
          if ((gc & 4) == 0) continue;  // its lower half duplicates the neighbour's columns
          gc -= chunk0 - (p.N - 8);
        }
        if (gc >= p.N) continue;
        float* dst = C32 + (size_t)gr * p.ldc + gc;
        f4 v = {alpha_ * acc[i][j][0] * post_, alpha_ * acc[i][j][1] * post_, alpha_ * acc[i][j][2] * post_, alpha_ * acc[i][j][3] * post_};
        if (p.beta != 0.0f) {
          const f4 o = *reinterpret_cast<const f4*>(dst);
          v = {v[0] + p.beta * o[0], v[1] + p.beta * o[1], v[2] + p.beta * o[2], v[3] + p.beta * o[3]};
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
      }
    return;
  }

  // ---- epilogue: lane holds C[row lane&15][cols 4*(lane>>4) .. +3] of each fragment.
  // N % 8 == 4 (the reference's column-major layout with m = 196): the B loads of the last, half-valid 8-column chunk
  // were clamped to columns N-8 .. N-1, so that chunk of the tile holds THOSE outputs: it is stored at N-8 as well
  // (the four columns it shares with its neighbour are written twice with the same values), or, element by element,
  // only its upper four columns.  A 16-byte store needs dword alignment only.
  const bool c_vec = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 7u) == 0);
  if (p.beta == 0.0f && c_vec) {
    char* Cs = smem;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const unsigned row = wm * TM + i * 16 + r, col = wn * TN + j * 16 + 4u * g;
        h4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = to_elt<BF>(p.alpha * acc[i][j][q]);
        *reinterpret_cast<h4*>(Cs + row * CPITCH + col * 2) = o;
      }
    __syncthreads();
    constexpr int C_CH = BM * (BN / 8) / (64 * NW);
#pragma unroll
    for (int i = 0; i < C_CH; ++i) {
      const unsigned q = tid + 64u * NW * i, row = q / (BN / 8), cn = q % (BN / 8);
      const int gr = m0 + (int)row;
      int gc = n0 + 8 * (int)cn;
      if (gr >= p.M || gc >= p.N) continue;
      gc = gc <= p.N - 8 ? gc : p.N - 8;
      __builtin_nontemporal_store(*reinterpret_cast<const u4*>(Cs + row * CPITCH + cn * 16), reinterpret_cast<u4*>(C + (size_t)gr * p.ldc + gc));  // see store_c_tile
    }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int gr = m0 + (int)(wm * TM + i * 16 + r);
        int gc = n0 + (int)(wn * TN + j * 16 + 4u * g);
        if (gr >= p.M) continue;
        const int chunk0 = gc & ~7;  // first column of this lane's 8-column chunk
        if (chunk0 < p.N && chunk0 > p.N - 8) {  // the clamped chunk: its lower half duplicates the neighbour's columns
          if ((gc & 4) == 0) continue;
          gc -= chunk0 - (p.N - 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (gc + q >= p.N) continue;
          half_t* dst = C + (size_t)gr * p.ldc + gc + q;
          float v = p.alpha * acc[i][j][q];
          if (p.beta != 0.0f) v += p.beta * to_f32<BF>(*dst);
          *dst = to_elt<BF>(v);
        }
      }
  }
}

template <int BM, int BN, int WM, int WN, int NS, bool BF = false, bool F32OUT = false>
static int launch_dma(const GemmArgs& a0, hipStream_t st) {
  GemmArgs a = a0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("gemm_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = NS * ((size_t)BM * 128 + (size_t)64 * BN * 2);
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  // fewer K stages than ring buffers: the unused buffers are not allocated (more workgroups per CU)
  const size_t used = ((size_t)(a.K / 64) < (size_t)NS ? (size_t)(a.K / 64) : (size_t)NS) * ((size_t)BM * 128 + (size_t)64 * BN * 2);
  const size_t lds_launch = used > lds_epi ? used : lds_epi;
  static LdsOptIn lds_optin;
  if (lds > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&gemm_f16_dma_kernel<BM, BN, WM, WN, NS, BF, F32OUT>), lds, "gemm_f16_dma_kernel")) return rc;
  }
  gemm_f16_dma_kernel<BM, BN, WM, WN, NS, BF, F32OUT><<<dim3((unsigned)nwg), dim3(64 * WM * WN), lds_launch, st>>>(a);
  return check_launch("gemm_f16_dma_kernel");
}

template <int BM, int BN, int WM, int WN, bool TA = false, bool TB = false, bool BF = false>
static int launch_cfg(const GemmArgs& a0, int vec, hipStream_t st) {
  GemmArgs a = a0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("gemm_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = (size_t)BM * 128 + (size_t)BN * 128;
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  if (vec == 8)
    gemm_f16_kernel<BM, BN, WM, WN, 8, TA, TB, BF><<<dim3((unsigned)nwg), dim3(256), lds, st>>>(a);
  else if (vec == 4 && !TA && !TB)
    gemm_f16_kernel<BM, BN, WM, WN, 4, false, false, BF><<<dim3((unsigned)nwg), dim3(256), lds, st>>>(a);
  else
    gemm_f16_kernel<BM, BN, WM, WN, 1, TA, TB, BF><<<dim3((unsigned)nwg), dim3(256), lds, st>>>(a);
  return check_launch("gemm_f16_kernel");
}

// Tile choice: the streamed dimension gets the long tile edge; narrow problems get a tile as wide
// as they are, so the big operand is read from HBM exactly once.
template <bool BF = false>
static int launch_gemm_f16(const GemmArgs& a, hipStream_t st, bool ta = false, bool tb = false, void* workspace = nullptr, size_t workspace_bytes = 0) {
  auto aligned_to = [&](unsigned halves) {
    const uintptr_t mask = halves * 2u - 1u;
    return (a.lda % halves == 0) && (a.ldb % halves == 0) && (a.sA % halves == 0) && (a.sB % halves == 0) &&
           (a.Ap || (reinterpret_cast<uintptr_t>(a.A) & mask) == 0) && (a.Bp || (reinterpret_cast<uintptr_t>(a.B) & mask) == 0);
  };
  const int vec = aligned_to(8) ? 8 : (aligned_to(4) ? 4 : 1);
  if constexpr (!BF) {  // transposed operands: batched::gemm only, which has no bfloat16 form
    if (ta && tb) return launch_cfg<128, 128, 2, 2, true, true>(a, vec, st);
    if (ta) return launch_cfg<128, 128, 2, 2, true, false>(a, vec, st);
    if (tb) return launch_cfg<128, 128, 2, 2, false, true>(a, vec, st);
  }
  // (round 4) the dense twin: ragged k and n > 128 run through the fused 2:4 kernels' pipelines with dense MFMA (spmma_f16_fused.hip:
  // gemm_dense_twin) where those are the better pipelines; SM_GEMM_TWIN (tuning): 0 = never, 2 = wherever supported
  if (a.ldb == a.N && a.ldc == a.N && a.M > 0) {
    const int twin_mode = tuning_int("SM_GEMM_TWIN", 1);
    if (twin_mode > 0) {
      DenseTwinCall c = {a.A, a.B, a.C, a.Ap, a.Bp, a.Cp, a.sA, a.sB, a.sC, a.M, a.N, a.K, a.lda, a.batch, a.alpha, a.beta, BF, twin_mode, workspace, workspace_bytes};
      const int rc = gemm_dense_twin(c, st);
      if (rc != SM_STATUS_NOT_SUPPORTED) return rc;
    }
  }
  // pointer-array batches: per-batch base alignment is the caller's (hipMalloc gives 256 B);
  // DMA fast path: whole 64-deep K stages, N a multiple of 4 (a half-valid last chunk is served from columns
  // N-8 .. N-1, see the kernel's epilogue), rows on 8-byte boundaries (a
  // 16-byte global access needs only dword alignment; the LDS side is aligned by construction)
  const bool fast = (a.K % 64 == 0) && (a.N % 4 == 0) && a.N >= 8 && (a.lda % 4 == 0) && (a.ldb % 4 == 0) &&
                    (a.sA % 4 == 0) && (a.sB % 4 == 0) && (a.Ap || (reinterpret_cast<uintptr_t>(a.A) & 7u) == 0) &&
                    (a.Bp || (reinterpret_cast<uintptr_t>(a.B) & 7u) == 0);
  if (fast) {
    if (a.N <= 64) return launch_dma<128, 64, 4, 1, 2, BF>(a, st);
    if (a.M <= 64) return launch_dma<64, 128, 1, 4, 2, BF>(a, st);
    return launch_dma<128, 128, 2, 2, 2, BF>(a, st);
  }
  if (a.N <= 64) return launch_cfg<128, 64, 4, 1, false, false, BF>(a, vec, st);
  if (a.M <= 64) return launch_cfg<64, 128, 1, 4, false, false, BF>(a, vec, st);
  return launch_cfg<128, 128, 2, 2, false, false, BF>(a, vec, st);
}

// fp16 x fp16 -> fp32 row-major product for spmm.hip's dense-MFMA COO form: C32[M x N] (ldc) = alpha * A[M x K] (lda) *
// (B[0 .. K) + B[K .. 2K)) (ldb, the two fp16 planes of one fp32 operand stacked along k) + beta * C32.  K % 64 == 0,
// N % 4 == 0, N >= 8, leading dimensions multiples of 4, 8-byte aligned bases (the DMA kernel's conditions).
int gemm_f16_f32out(const void* A, const void* B2, float* C, size_t M, size_t N, size_t K, size_t lda, size_t ldb, size_t ldc, float alpha,
                    float beta, hipStream_t st, const float* alpha_dev, const int* skip_flag) {
  if (K % 64 != 0 || N % 4 != 0 || N < 8 || lda % 4 != 0 || ldb % 4 != 0 || ldc % 4 != 0 || M > 0x7fffffffull || N > 0x7fffffffull ||
      2 * K > 0x7fffffffull || (reinterpret_cast<uintptr_t>(A) & 7u) || (reinterpret_cast<uintptr_t>(B2) & 7u) || (reinterpret_cast<uintptr_t>(C) & 15u)) {
    set_error("gemm_f16_f32out: shape / alignment not served");
    return SM_STATUS_NOT_SUPPORTED;
  }
  GemmArgs a = {};
  a.A = (const half_t*)A; a.B = (const half_t*)B2; a.C32 = C;
  a.M = (int)M; a.N = (int)N; a.K = (int)(2 * K);
  a.lda = (int)lda; a.ldb = (int)ldb; a.ldc = (int)ldc;
  a.batch = 1; a.alpha = alpha; a.beta = beta;
  a.a_wrap_kt = (int)(K / 64);
  a.alpha_dev = alpha_dev;
  a.skip_flag = skip_flag;
  if (N <= 64) return launch_dma<128, 64, 4, 1, 2, false, true>(a, st);
#ifdef SM_TUNING
  switch (tuning_int("SM_COOFAST_TILE", 0)) {  // A/B of the tile shape of this compute-bound product
    case 1: return launch_dma<256, 128, 4, 2, 2, false, true>(a, st);
    case 2: return launch_dma<128, 256, 2, 4, 2, false, true>(a, st);
    case 3: return launch_dma<256, 128, 4, 2, 3, false, true>(a, st);
    case 4: return launch_dma<128, 128, 2, 2, 3, false, true>(a, st);
    case 5: return launch_dma<256, 256, 4, 4, 2, false, true>(a, st);
    case 6: return launch_dma<64, 128, 1, 4, 2, false, true>(a, st);
    case 7: return launch_dma<128, 64, 4, 1, 2, false, true>(a, st);
    case 8: return launch_dma<64, 128, 1, 4, 3, false, true>(a, st);
    case 9: return launch_dma<256, 256, 2, 4, 2, false, true>(a, st);
    case 10: return launch_dma<256, 128, 2, 2, 2, false, true>(a, st);
    case 11: return launch_dma<128, 256, 2, 2, 2, false, true>(a, st);
    default: break;
  }
#endif
  // 256 x 256 tiles (16 waves, one workgroup per CU) halve the LDS bytes per flop of this compute-bound product; taken when they
  // still give most CUs a tile and pad N no worse than 128-wide tiles do (profiles/coo_fast_tiles_r03p.txt: 137 vs 157 us on
  // (2048 x 12544 x 1152), 101 vs 109 on (4096 x 3136 x 2304); 276 vs 199 and 148 vs 116 where the conditions fail)
  {
    const size_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
    const double pad256 = (double)((N + 255) / 256 * 256) / (double)N, pad128 = (double)((N + 127) / 128 * 128) / (double)N;
    if (t256 >= 192 && pad256 <= 1.1 * pad128) return launch_dma<256, 256, 4, 4, 2, false, true>(a, st);
    // at most one 128 x 128 tile per CU: 64-row tiles put two workgroups on a CU (16384 x 196 x 9216: 181 vs 202 us)
    const size_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
    if (t128 <= (size_t)device_cu_count()) return launch_dma<64, 128, 1, 4, 2, false, true>(a, st);
  }
  return launch_dma<128, 128, 2, 2, 2, false, true>(a, st);
}

}  // namespace sm

using namespace sm;

extern "C" {

}  // extern "C"

template <bool BF>
static int gemm_rowmajor16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                           size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                           sm_stream_t stream, void* workspace = nullptr, size_t workspace_bytes = 0) {
  if (!A || !B || !C || lda < k) {
    set_error("sm_gemm_rowmajor_{f16,bf16}: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || lda > 0x7fffffffull) {
    set_error("sm_gemm_rowmajor_{f16,bf16}: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  GemmArgs a = {};
  a.A = (const half_t*)A; a.B = (const half_t*)B; a.C = (half_t*)C;
  a.M = (int)m; a.N = (int)n; a.K = (int)k;
  a.lda = (int)lda; a.ldb = (int)n; a.ldc = (int)n;
  a.sA = strideA; a.sB = strideB; a.sC = strideC;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  // a batch that shares B and stacks A and C contiguously is one tall matrix: no per-batch tails
  if (batch > 1 && strideB == 0 && strideA == m * lda && strideC == m * n) {
    a.M = (int)(m * batch);
    a.batch = 1;
  }
  return launch_gemm_f16<BF>(a, (hipStream_t)stream, false, false, workspace, workspace_bytes);
}

extern "C" {

int sm_gemm_rowmajor_f16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                         size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                         sm_stream_t stream) {
  return gemm_rowmajor16<false>(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream);
}
/* bfloat16 form of the same kernels (v_mfma_f32_16x16x32_bf16): the dense denominator of sm_spmma_bf16 */
int sm_gemm_rowmajor_bf16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                          size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                          sm_stream_t stream) {
  return gemm_rowmajor16<true>(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream);
}

static int gemm_batched16(const void* const* A_ptrs, const void* const* B_ptrs, void* const* C_ptrs, size_t m,
                          size_t n, size_t k, size_t batch, int ta, int tb, float alpha, float beta,
                          sm_stream_t stream, void* workspace, size_t workspace_bytes) {
  if (!A_ptrs || !B_ptrs || !C_ptrs) {
    set_error("sm_gemm_batched_f16: null pointer array");
    return SM_STATUS_INVALID_VALUE;
  }
  if ((ta != SM_OP_N && ta != SM_OP_T) || (tb != SM_OP_N && tb != SM_OP_T)) {
    set_error("sm_gemm_batched_f16: operation must be SM_OP_N or SM_OP_T");
    return SM_STATUS_INVALID_VALUE;
  }
  // the reference passes lda = m, ldb = k whatever the flags (gemm.hxx:80-81); a transposed operand is then read as
  // its k x m (n x k) stored form through that same leading dimension, which the vendor BLAS accepts only when it
  // covers a stored column: lda >= k for op(A) = T, ldb >= n for op(B) = T
  if ((ta == SM_OP_T && m < k) || (tb == SM_OP_T && k < n)) {
    set_error("sm_gemm_batched_f16: leading dimension (lda = m, ldb = k) shorter than a column of the transposed operand");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || batch > 0x7fffffffull) {
    set_error("sm_gemm_batched_f16: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  // column-major C = A*B  <=>  row-major C^T[n x m] = B^T[n x k] * A^T[k x m]
  GemmArgs a = {};
  a.Ap = (const half_t* const*)B_ptrs;  // "A" of the row-major product is the reference's B
  a.Bp = (const half_t* const*)A_ptrs;
  a.Cp = (half_t* const*)C_ptrs;
  a.M = (int)n; a.N = (int)m; a.K = (int)k;
  a.lda = (int)k; a.ldb = (int)m; a.ldc = (int)m;
  a.batch = (int)batch; a.alpha = alpha; a.beta = beta;
  // op(B) = T: the row-major view's A operand (n x k) arrives n-contiguous; op(A) = T: its B operand (k x m) arrives
  // k-contiguous.  The leading dimensions stay (ldb = k, lda = m), as the reference passes them.
  return launch_gemm_f16(a, (hipStream_t)stream, tb == SM_OP_T, ta == SM_OP_T, workspace, workspace_bytes);
}
int sm_gemm_batched_f16(const void* const* A_ptrs, const void* const* B_ptrs, void* const* C_ptrs, size_t m,
                        size_t n, size_t k, size_t batch, int ta, int tb, float alpha, float beta,
                        sm_stream_t stream) {
  return gemm_batched16(A_ptrs, B_ptrs, C_ptrs, m, n, k, batch, ta, tb, alpha, beta, stream, nullptr, 0);
}
/* the dense entry points with the fused kernels' stream-K workspace (round 5): the dense twin of sm_spmma_fused_*_ws */
int sm_gemm_batched_f16_ws(const void* const* A_ptrs, const void* const* B_ptrs, void* const* C_ptrs, size_t m,
                           size_t n, size_t k, size_t batch, int ta, int tb, float alpha, float beta,
                           void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  return gemm_batched16(A_ptrs, B_ptrs, C_ptrs, m, n, k, batch, ta, tb, alpha, beta, stream, workspace, workspace_bytes);
}
int sm_gemm_rowmajor_f16_ws(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                            size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                            void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  return gemm_rowmajor16<false>(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream, workspace, workspace_bytes);
}
int sm_gemm_rowmajor_bf16_ws(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                             size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                             void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  return gemm_rowmajor16<true>(A, B, C, m, n, k, lda, batch, strideA, strideB, strideC, alpha, beta, stream, workspace, workspace_bytes);
}

}  // extern "C"
