// spmma_f16_pc.hip -- producer/consumer 2:4 matmul with 128-deep stages and big tiles.
//
// Why this shape (measured, profiles/stamp_r01.txt, tools/fillrate.hip): a CU ingests at most ~62 B/clk from
// L2 and an LDS-DMA instruction costs time in proportion to the 128-byte cache lines it touches, so the
// kernel is bound by LINES INGESTED PER MAC.  Relative to the 64-deep, 128 x 128 kernel of spmma_f16.hip
// (A in 64-byte half lines, B re-fetched per 128 rows): a 128-deep stage moves A in whole lines, and a
// 256-row tile amortises every B line over twice the rows: 136 instead of 264 lines per 128x128x64 MACs.
//
// Stage = 128 dense k: A values [BM][128 B] (chunk c of row r at c ^ (r & 7)), metadata two planes
// [2][BM][8 B] (the stage-major blob layout makes each plane of a tile contiguous), B [BN/64][128][128 B]
// (chunk swizzle b_swz).  NL loader waves issue all DMA; WM x WN consumer waves read LDS and issue
// v_smfmac_f32_16x16x64_f16 (two k-steps per stage); one s_barrier per stage joins them (spmma_f16.hip
// explains the protocol).  Edge rows / columns / k-rows / metadata planes are clamped, never predicated,
// so every loader wave issues the same number of DMA instructions in every stage (the counted vmcnt
// depends on it); a trailing half stage (kc % 128 == 64) simply skips its second k-step.
#include "spmma_args.h"

namespace sm {

template <int BM, int BN, int WM, int WN, int NL, int NS>
__global__ __launch_bounds__(64 * (WM * WN + NL)) void spmma_f16_pc2_kernel(const SpmmaArgs p) {
  constexpr int NC = WM * WN, NW = NC + NL;
  static_assert(NS >= 2 && NS <= 3, "ring depth");
  static_assert(BM % 128 == 0 && BN % 64 == 0, "tile");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(FM >= 1 && FN >= 1, "wave tile");
  constexpr int SA = BM * 128, SMP = BM * 8, SB = 128 * BN * 2, STAGE = SA + 2 * SMP + SB;
  constexpr int A_N = BM / 8, M_N = 2 * (BM / 128), B_N = BN / 4, W = A_N + M_N + B_N;
  constexpr int SL = (W + NL - 1) / NL, LPS = W / NL;
  static_assert(W % NL == 0, "every loader wave issues the same number of DMA instructions per stage");
  constexpr int CPITCH = BN * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = lid / tiles, trem = lid - b * tiles;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = (p.kc + 127) / 128;
  half_t* C = p.C + (size_t)b * p.sC;

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned g = lane >> 4, r = lane & 15u;
  const unsigned wm = wave / WN, wn = wave % WN;  // consumer waves only

  if (wave >= (unsigned)NC) {
    // ------------------------------------------------------------------ loader wave
    const unsigned lw = wave - NC;
    const size_t row_base = (size_t)b * p.m;
    const char* vals = p.vals + row_base * (size_t)p.kc;
    const char* meta = p.meta + row_base * 8;
    const char* Bb = reinterpret_cast<const char*>(p.B + (size_t)b * p.sB);
    const int mlast = p.Mrows - 1;
    const int nplanes = p.kc / 64;
    // per slot: A -> base = row pointer (+128 B per stage); metadata -> base = plane-0 pointer of this
    // lane's row pair, aux = which plane of the stage; B -> base = column pointer in row 0, aux = k-row
    const char* base[SL];
    unsigned aux[SL], loff[SL];
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const unsigned t = lw + (unsigned)NL * i;
      if (t < (unsigned)A_N) {
        const unsigned row = 8u * t + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
        int gr = m0 + (int)row;
        gr = gr < mlast ? gr : mlast;
        base[i] = vals + (size_t)gr * p.kc + 16u * cs;
        aux[i] = 0;
        loff[i] = t * 1024u;
      } else if (t < (unsigned)(A_N + M_N)) {
        const unsigned u = t - A_N, pl = u / (BM / 128), blk = u % (BM / 128);
        size_t off = ((size_t)m0 + 128u * blk + 2u * lane) * 8;
        const size_t last = (size_t)p.Mrows * 8 - 16;
        off = off < last ? off : last;
        base[i] = meta + off;
        aux[i] = pl;
        loff[i] = SA + pl * SMP + blk * 1024u;
      } else {
        const unsigned j = t - (A_N + M_N), panel = j >> 4, kr = 8u * (j & 15u) + (lane >> 3);
        const unsigned cs = (lane & 7u) ^ b_swz(kr);
        int gc = n0 + (int)(64u * panel + 8u * cs);
        gc = gc <= p.N - 8 ? gc : p.N - 8;
        base[i] = Bb + (size_t)gc * 2;
        aux[i] = kr;
        loff[i] = SA + 2 * SMP + panel * (128u * 128u) + (j & 15u) * 1024u;
      }
    }
    auto stage = [&](int kt, int buf) {
      char* sb = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < SL; ++i) {
        const unsigned t = lw + (unsigned)NL * i;  // wave-uniform
        const char* src;
        if (t < (unsigned)A_N) {
          src = base[i] + (size_t)kt * 128;
        } else if (t < (unsigned)(A_N + M_N)) {
          int plane = 2 * kt + (int)aux[i];
          plane = plane < nplanes ? plane : nplanes - 1;
          src = base[i] + (size_t)plane * p.Mtot * 8;
        } else {
          int gk = kt * 128 + (int)aux[i];
          gk = gk < p.K ? gk : p.K - 1;
          src = base[i] + (size_t)gk * p.N * 2;
        }
        __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(sb + loff[i]), 16, 0, 0);
      }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nkt) stage(s, s);
    int fill = NS - 1;
    for (int kt = 0; kt < nkt; ++kt) {
      const int ahead = (nkt - 1 - kt) < (NS - 2) ? (nkt - 1 - kt) : (NS - 2);
      if (NS >= 3 && ahead == 1) wait_dma_and_barrier<LPS>();
      else wait_dma_and_barrier<0>();
      if (kt + NS - 1 < nkt) stage(kt + NS - 1, fill);
      fill = fill + 1 == NS ? 0 : fill + 1;
    }
  } else {
    // ------------------------------------------------------------------ consumer wave
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      wait_dma_and_barrier<0>();  // no DMA of its own: this is the stage barrier
      const char* As = smem + cur * STAGE;
      const char* Ms = As + SA;
      const char* Bs = Ms + 2 * SMP;
      const unsigned bs_addr = (unsigned)(uintptr_t)(lds_char*)Bs;
      const int nstep = (p.kc - kt * 128) >= 128 ? 2 : 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s >= nstep) break;
        h8 af[FM];
        int idx[FM];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const unsigned row = wm * TM + i * 16 + r;
          af[i] = *reinterpret_cast<const h8*>(As + a_off(row, 4u * s + g));
          idx[i] = (int)*reinterpret_cast<const unsigned short*>(Ms + s * SMP + row * 8u + 2u * g);
        }
        s4 t0[2], t1[2], t2[2], t3[2];
        auto issue = [&](int j, s4& v0, s4& v1, s4& v2, s4& v3) {
          const unsigned col0 = wn * TN + j * 16, q = r >> 2, pp = r & 3u;
          const unsigned a = bs_addr + b_off<128>(64u * s + 8u * g + q, col0 + 4u * pp);
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
          for (int i = 0; i < FM; ++i)
            acc[i][j] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(af[i], bf, acc[i][j], idx[i], 0, 0);
        }
      }
      cur = cur + 1 == NS ? 0 : cur + 1;
    }
  }
  __syncthreads();  // both roles; nothing is in flight (the last NS-1 loader iterations issued no DMA)

  // ---- epilogue: consumers stage their fragments, every wave stores 16-byte row pieces
  const bool c_vec = (reinterpret_cast<uintptr_t>(C) & 15u) == 0;
  if (p.beta == 0.0f && c_vec) {
    char* Cs = smem;
    if (wave < (unsigned)NC) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const unsigned row = wm * TM + i * 16 + 4u * g, col = wn * TN + j * 16 + r;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<half_t*>(Cs + (row + q) * CPITCH + col * 2) = (half_t)(p.alpha * acc[i][j][q]);
        }
    }
    __syncthreads();
    constexpr int NCH = BM * (BN / 8);
    for (unsigned q = tid; q < (unsigned)NCH; q += 64u * NW) {
      const unsigned row = q / (BN / 8), cn = q % (BN / 8);
      const int gr = m0 + (int)row, gc = n0 + 8 * (int)cn;
      if (gr >= p.Mrows || gc >= p.N) continue;
      *reinterpret_cast<u4*>(C + (size_t)gr * p.N + gc) = *reinterpret_cast<const u4*>(Cs + row * CPITCH + cn * 16);
    }
  } else if (wave < (unsigned)NC) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int gc = n0 + (int)(wn * TN + j * 16 + r);
        if (gc >= p.N) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gr = m0 + (int)(wm * TM + i * 16 + 4u * g) + q;
          if (gr >= p.Mrows) continue;
          half_t* dst = C + (size_t)gr * p.N + gc;
          float v = p.alpha * acc[i][j][q];
          if (p.beta != 0.0f) v += p.beta * (float)*dst;
          *dst = (half_t)v;
        }
      }
  }
}

template <int BM, int BN, int WM, int WN, int NL, int NS>
static int launch_pc2(const SpmmaArgs& a0, hipStream_t st) {
  SpmmaArgs a = a0;
  a.tiles_m = (a.Mrows + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("spmma_f16: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = NS * ((size_t)BM * 144 + (size_t)128 * BN * 2);
  constexpr size_t lds_epi = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spmma_f16_pc2_kernel<BM, BN, WM, WN, NL, NS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  spmma_f16_pc2_kernel<BM, BN, WM, WN, NL, NS><<<dim3((unsigned)nwg), dim3(64 * (WM * WN + NL)), lds, st>>>(a);
  return check_launch("spmma_f16_pc2_kernel");
}

int spmma_f16_pc2_launch(const SpmmaArgs& a, int cfg, int ns, hipStream_t st) {
  switch (cfg) {
    case 1:  // 256 x 128, 8 consumer waves (64 x 64 each) + 4 loaders; W = 32 + 4 + 32 = 68
      return launch_pc2<256, 128, 4, 2, 4, 2>(a, st);
    case 2:  // 128 x 64, 4 consumer waves (32 x 64) + 2 loaders; W = 16 + 2 + 16 = 34
      return ns >= 3 ? launch_pc2<128, 64, 4, 1, 2, 3>(a, st) : launch_pc2<128, 64, 4, 1, 2, 2>(a, st);
    default:  // 128 x 128, 4 consumer waves (64 x 64) + 2 loaders; W = 16 + 2 + 32 = 50
      return ns >= 3 ? launch_pc2<128, 128, 2, 2, 2, 3>(a, st) : launch_pc2<128, 128, 2, 2, 2, 2>(a, st);
  }
}

}  // namespace sm
