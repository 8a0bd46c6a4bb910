// spmm_coo_smfmac.hip -- config 5 (sparsifyme::batched::strided_coo, reference include/sparsify.me/spmm.hxx:140-187; the workload of
// profiling/python/gemm_coo_compare.py:7: A 10 % dense) on the SPARSE matrix instruction; what sm_spmm_coo_f32_fast runs for beta == 0
// and a density of at most 20 % (round 5; the dense-MFMA pipeline of spmm.hip for everything else).
//
// All batches are one product: C [m x NV] = alpha * A [m x k] * B [k x NV], NV = n * batches, B and C column-major with leading
// dimensions k and m (the batch strides are exactly k n and m n).  A random 10 %-dense A violates 2:4 in 0.4 % of its 1 x 4 strips, so:
//   * `coo_smfmac_compress_kernel` (one workgroup per 128-row x 64-k block of the dense fp32 scatter of A, duplicates added): per strip
//     the first two non-zeros go to a 2:4 image -- values scaled by the call's power of two and split hi + lo into two fp16 planes
//     (|error| <= 2^-22 |a|), ONE index nibble for both -- and a strip's third / fourth non-zero go to the bucket of their 32-row block:
//     64 slots of {row, k, scaled fp32 value}, filled in a fixed order (a wave-wide prefix sum: no atomics, reproducible), the rest marked
//     empty, so that the matrix kernel fetches a bucket with one unconditional load per lane;
//   * `spmm_coo_smfmac_kernel`: a 128 x 128 tile of C per workgroup, four waves of 32 rows x 128 columns.  Per 64-k stage the fp32
//     columns of B come in with 16-byte loads (a column's 64 k are 256 contiguous bytes), are scaled and rounded to fp16 on the way
//     into an LDS image [column][64 k] -- column-major B IS the operand order of v_smfmac_f32_16x16x64_f16's dense side: two
//     ds_read_b128 per fragment, no transposing reads -- the 2:4 fragments of A come straight from L2 into the lanes that feed them
//     (a 16-row fragment of a stage is 1 KiB contiguous), two SMFMACs per fragment pair (hi, lo), then the wave's bucket entries (prefetched
//     a stage ahead, one per lane, handed out with v_readlane) as fp32 multiply-adds of the entry's value with the fp16 row of B in LDS, into
//     the accumulator register the SMFMAC result map assigns to that row.  The result map leaves four consecutive rows of one column in a lane = 16 contiguous bytes of column-major C; the tile
//     goes through LDS once so that every wave store writes two whole 512-byte column pieces.
// Error: the dense operand is rounded once to fp16 (2^-11 relative per element), A is exact to 2^-22 (image) or exact (bucket entries):
// the bound of the dense-MFMA form.  Range: an element of B the scale cannot bring into fp16 raises the header's flag and the tile
// that met it is not stored (beta == 0: the caller recomputes C by the exact form anyway); an A value out of range or a bucket beyond
// its capacity (a 32 x 64 block with more than 64 third / fourth non-zeros: far denser than this form is meant for) raises it before the
// matrix kernel starts, which then returns at once.
// Measured (profiles/coo_config5_r05am.txt, coo_ablate_r05ac.txt, coo_ablate_r05al.txt): 45-165 us per call over config 5's shapes = 0.06-0.51 of the HBM roofline;
// the call's scan / scatter / image kernels are 25-45 us of that, and a stage of the matrix kernel costs one load latency: B must pass through
// registers to become fp16, so its loads are one stage ahead at most (four or eight waves per tile: the same times).
#include <type_traits>

#include "coo_fast.h"
#include "mma_tile.h"

namespace sm {

constexpr int CS_SEG = 64;   // bucket slots per (32 rows x 64 k) block of A
constexpr int CS_CAP = 4 * CS_SEG;
constexpr unsigned CS_EMPTY = 0xffffffffu;
#ifdef SM_TUNING
#define SM_COO_ABL(bit) ((p.ablate & (bit)) != 0)
#else
#define SM_COO_ABL(bit) false
#endif

struct CooSmArgs {
  const float* A32;       // dense fp32 scatter of A, [m][kc] (zero padded to whole stages)
  half_t* hi;             // [nst][m][32] kept values, high parts
  half_t* lo;             // same, low parts
  unsigned short* meta;   // [nst][m][4]: one index halfword per (row, 16-k group)
  u2* rlist;              // [tiles_m][nst][4][CS_SEG] {row in tile | k in stage << 8, scaled value bits}; CS_EMPTY in an unused slot
  const float* B;
  float* C;
  CooFastHdr* hdr;
  int m, k, kc, nst, tiles_m, tiles_nv, slice_w;
  long long nv;
  float alpha;
  int ablate;  // diagnostic timing builds (-DSM_TUNING, SM_COO_ABLATE): 1 no bucket entries, 2 no loads of B, 4 no SMFMAC; 0 in the product
};

__global__ __launch_bounds__(256) void coo_scatter_rows_kernel(const int* __restrict__ rows, const int* __restrict__ cols, const float* __restrict__ vals, size_t nnz,
                                                               size_t m, size_t k, size_t kc, float* __restrict__ A32) {
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < nnz; e += (size_t)gridDim.x * 256) {
    const size_t r = (size_t)rows[e], c = (size_t)cols[e];
    if (r < m && c < k) atomicAdd(A32 + r * kc + c, vals[e]);  // an out-of-range coordinate is skipped, as in the other COO forms
  }
}

__global__ __launch_bounds__(256) void coo_smfmac_compress_kernel(const CooSmArgs p) {
  const unsigned tile = blockIdx.x / (unsigned)p.nst, s = blockIdx.x - tile * (unsigned)p.nst;
  const unsigned t = threadIdx.x, rt = t >> 1, half = t & 1u, lane = t & 63u, wave = t >> 6;
  const int row = (int)tile * 128 + (int)rt;
  const bool rok = row < p.m;
  const float sc = coo_fast_pow2(coo_fast_scale_exp(p.hdr->max_a, 13));
  bool bad = false;
  f4 x[2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (rok) v = *reinterpret_cast<const f4*>(p.A32 + (size_t)row * p.kc + s * 64u + (2u * half + c) * 16u + 4u * q);
      x[c][q] = v * sc;
    }
  unsigned ext[2][4];  // per strip: mask of the positions that go to the bucket
  int ne = 0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    uint32_t hw[4], lw[4];
    unsigned idx = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 a = x[c][q];
      bad |= coo_fast_out_of_range(a[0]) || coo_fast_out_of_range(a[1]) || coo_fast_out_of_range(a[2]) || coo_fast_out_of_range(a[3]);
      const unsigned nz = (a[0] != 0.f ? 1u : 0u) | (a[1] != 0.f ? 2u : 0u) | (a[2] != 0.f ? 4u : 0u) | (a[3] != 0.f ? 8u : 0u);
      const unsigned cnt = (unsigned)__builtin_popcount(nz);
      const unsigned first = nz ? (unsigned)__builtin_ctz(nz) : 0u, rest = nz & (nz - 1u);
      const unsigned second = rest ? (unsigned)__builtin_ctz(rest) : 3u;
      ext[c][q] = rest & (rest - 1u);
      ne += (int)__builtin_popcount(ext[c][q]);
      // kept pair p0 < p1 covering the first two non-zeros (fewer: any other position, whose value is its true zero)
      const unsigned p0 = cnt >= 2u ? first : (cnt == 1u && first < 3u ? first : 0u);
      const unsigned p1 = cnt >= 2u ? second : 3u;
      float v0 = a[0], v1 = a[1];
      v0 = p0 == 1u ? a[1] : v0;
      v0 = p0 == 2u ? a[2] : v0;
      v1 = p1 == 2u ? a[2] : v1;
      v1 = p1 == 3u ? a[3] : v1;
      const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
      const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
      hw[q] = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
      lw[q] = (uint32_t)__builtin_bit_cast(unsigned short, l0) | ((uint32_t)__builtin_bit_cast(unsigned short, l1) << 16);
      idx |= (p0 | (p1 << 2)) << (4 * q);
    }
    if (rok) {
      const size_t o = ((size_t)s * p.m + row) * 32 + (2u * half + c) * 8u;
      *reinterpret_cast<u4*>(p.hi + o) = u4{hw[0], hw[1], hw[2], hw[3]};
      *reinterpret_cast<u4*>(p.lo + o) = u4{lw[0], lw[1], lw[2], lw[3]};
      p.meta[((size_t)s * p.m + row) * 4 + 2u * half + c] = (unsigned short)idx;
    }
  }
  // ---- the bucket: one segment of CS_SEG slots per 32-row block (= per wave here, = per wave or wave pair of the matrix kernel), entries in a
  //      fixed order (row, group, strip, position) by a wave-wide prefix sum of the counts, the unused slots marked empty: the matrix kernel
  //      brings a segment in with ONE unconditional load per lane and counts the valid slots itself (no count to fetch first)
  int incl = ne;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(incl, o);
    incl += (int)lane >= o ? y : 0;
  }
  const int total = __builtin_amdgcn_readlane(incl, 63);
  int base = incl - ne;
  u2* dst = p.rlist + (((size_t)tile * p.nst + s) * 4 + wave) * CS_SEG;
  if ((int)lane >= total) dst[lane] = u2{CS_EMPTY, 0u};
  if (total > CS_SEG && lane == 0u) atomicOr(&p.hdr->flag, 4);
  if (ne) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q)
      {  // a third / fourth non-zero sits at position 2 or 3 (named copies: __builtin_bit_cast applied to a vector ELEMENT expression reads
         // element 0 with this compiler)
        const unsigned kq = (2u * half + c) * 16u + 4u * q;
        const float xz = x[c][q][2], xw = x[c][q][3];
        if (ext[c][q] & 4u) {
          if (base < CS_SEG) dst[base] = u2{rt | ((kq + 2u) << 8), __builtin_bit_cast(uint32_t, xz)};
          ++base;
        }
        if (ext[c][q] & 8u) {
          if (base < CS_SEG) dst[base] = u2{rt | ((kq + 3u) << 8), __builtin_bit_cast(uint32_t, xw)};
          ++base;
        }
      }
  }
  if (__any(bad) && lane == 0u) atomicOr(&p.hdr->flag, 2);
}

template <bool KVEC, int NW>  // KVEC: k % 4 == 0, whole 16-byte pieces of B's columns; NW = 4 or 8 waves of 32 or 16 rows
__global__ __launch_bounds__(64 * NW, NW / 2) void spmm_coo_smfmac_kernel(const CooSmArgs p) {
  constexpr int BM = 128, BN = 128, TM = BM / NW, FM = TM / 16, FN = 8, SB = BN * 128, CP = BM * 4 + 16, NT = 64 * NW, BL = 2048 / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // main loop: [2][BN columns][64 k fp16]; epilogue: [BN][CP]
  __shared__ int wg_bad;
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (__builtin_amdgcn_readfirstlane(p.hdr->flag) != 0) return;  // A left the range / a bucket overflowed: C untouched, the caller falls back
  if (tid == 0) wg_bad = 0;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  // tile order: slices of `slice_w` column tiles; inside a slice the column tiles of one row tile follow one another.  An XCD (a contiguous
  // range of logical ids) then keeps its slice of B in L2 while A's row tiles stream through once per slice -- with the row tiles fastest
  // over ALL column tiles, the whole image of A (16 MB at 12544 x 576) missed L2 once per column tile: 256 MB for a 107 MB product
  const unsigned per_slice = (unsigned)p.tiles_m * (unsigned)p.slice_w;
  const unsigned sl = lid / per_slice, rem = lid - sl * per_slice;
  const unsigned width = (sl + 1u) * (unsigned)p.slice_w <= (unsigned)p.tiles_nv ? (unsigned)p.slice_w : (unsigned)p.tiles_nv - sl * (unsigned)p.slice_w;
  const unsigned tile_m = rem / width, tile_nv = sl * (unsigned)p.slice_w + (rem - tile_m * width);
  const int m0 = (int)tile_m * BM;
  const long long n0 = (long long)tile_nv * BN;
  const int xb = coo_fast_scale_exp(p.hdr->max_b, 12), xa = coo_fast_scale_exp(p.hdr->max_a, 13);
  const float scb = coo_fast_pow2(xb);

  // ---- B loader: lane -> (column wave * 32 + 4 i + (lane >> 4), 16-byte piece lane & 15 of the column's 64-k stage)
  const unsigned piece = lane & 15u, csub = lane >> 4;
  const float* bsrc[BL];
  bool bon[BL];
  unsigned bdst[BL];
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    const unsigned col = wave * (4u * BL) + 4u * i + csub;
    bon[i] = n0 + col < p.nv;
    bsrc[i] = p.B + (size_t)(bon[i] ? n0 + col : 0) * p.k + 4u * piece;
    bdst[i] = a_off(col, piece >> 1) + 8u * (piece & 1u);
  }
  f4 breg[BL];
  auto load_b = [&](int s) {
    const int kk = SM_COO_ABL(2) ? p.k : s * 64 + 4 * (int)piece;
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if constexpr (KVEC) {
        if (bon[i] && kk < p.k) v = *reinterpret_cast<const f4*>(bsrc[i] + s * 64);  // re-read by the other row tiles: no non-temporal hint
      } else {
        if (bon[i]) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (kk + e < p.k) v[e] = bsrc[i][s * 64 + e];
        }
      }
      breg[i] = v;
    }
  };
  bool bad = false;
  auto store_b = [&](int buf) {
    char* Bs = smem + buf * SB;
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const f4 x = breg[i] * scb;
      bad |= coo_fast_out_of_range(x[0]) || coo_fast_out_of_range(x[1]) || coo_fast_out_of_range(x[2]) || coo_fast_out_of_range(x[3]);
      typedef _Float16 hv4 __attribute__((ext_vector_type(4)));
      const hv4 h = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
      *reinterpret_cast<hv4*>(Bs + bdst[i]) = h;
    }
  };

  // ---- A fragments straight from the image: lane (row l & 15 of the fragment, 16-k group l >> 4)
  const unsigned g = lane >> 4, r16 = lane & 15u;
  size_t aoff[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int row = m0 + (int)wave * TM + 16 * i + (int)r16;
    row = row < p.m ? row : p.m - 1;
    aoff[i] = (size_t)row * 4 + g;  // in 16-byte pieces of a [m][64 B] stage plane; metadata halfword index is the same number
  }
  u4 ah[2][FM], al[2][FM];
  unsigned short am[2][FM];
  auto load_a = [&](int s, int slot) {
    const size_t so = (size_t)s * p.m * 4;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      ah[slot][i] = *(reinterpret_cast<const u4*>(p.hi) + so + aoff[i]);
      al[slot][i] = *(reinterpret_cast<const u4*>(p.lo) + so + aoff[i]);
      am[slot][i] = p.meta[so + aoff[i]];
    }
  };

  // ---- this wave's bucket entries of a stage (rows 32 wave .. + 31 of the tile: a contiguous run of the bucket), one per lane
  const int nst = p.nst;
  u2 rent[2];
  auto load_r = [&](int s, int slot) {
    const unsigned blk = wave * (unsigned)TM / 32u;  // the wave's 32-row block (NW = 8: two waves share one and skip each other's rows)
    rent[slot] = p.rlist[(((size_t)tile_m * nst + s) * 4 + blk) * CS_SEG + lane];
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  load_b(0);
  load_a(0, 0);
  load_r(0, 0);
  store_b(0);
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const int cur = s & 1;
    if (s + 1 < nst) {
      load_b(s + 1);
      load_a(s + 1, cur ^ 1);
      load_r(s + 1, cur ^ 1);
    }
    const char* Bs = smem + cur * SB;
    if (!SM_COO_ABL(4))
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const unsigned col = 16u * j + r16;
      const u4 b0 = *reinterpret_cast<const u4*>(Bs + a_off(col, g));
      const u4 b1 = *reinterpret_cast<const u4*>(Bs + a_off(col, 4u + g));
      typedef uint32_t u8v __attribute__((ext_vector_type(8)));
      const h16 bf = __builtin_bit_cast(h16, u8v{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]});
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        acc[i][j] = smfmac16<false>(__builtin_bit_cast(h8, ah[cur][i]), bf, acc[i][j], (int)am[cur][i]);
        acc[i][j] = smfmac16<false>(__builtin_bit_cast(h8, al[cur][i]), bf, acc[i][j], (int)am[cur][i]);
      }
    }
    // ---- the wave's third / fourth non-zeros of the stage: fp32 value x the fp16 row of B, into the register the result map gives the row
    //      (entries come out of the prefetched lane registers with v_readlane: no memory access inside the loop)
    {
      const u2 my = rent[cur];
      const int ne = SM_COO_ABL(1) ? 0 : (int)__builtin_popcountll(__ballot(my[0] != CS_EMPTY));  // the valid slots are a prefix of the segment
      for (int e = 0; e < ne; ++e) {
        const unsigned pos = (unsigned)__builtin_amdgcn_readlane((int)my[0], e);
        const int vbits = __builtin_amdgcn_readlane((int)my[1], e);
        const float v = __builtin_bit_cast(float, vbits);
        const unsigned rr = pos & 127u, kk = pos >> 8;
        if (FM == 1 && (rr >> 4) != wave) continue;  // the other wave of the 32-row block
        const float vs = ((rr >> 2) & 3u) == g ? v : 0.f;
        float bj[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j)
          bj[j] = (float)*reinterpret_cast<const _Float16*>(Bs + a_off(16u * j + r16, kk >> 3) + 2u * (kk & 7u));
        const unsigned sel = FM == 1 ? (rr & 3u) : ((rr >> 4) & 1u) * 4u + (rr & 3u);
        // (a chain of uniform ifs, each updating its registers in place: a switch made the compiler copy all 64 accumulators around every entry)
#define SM_COO_CASE(S)                                                                                                         \
  if ((S >> 2) < FM && sel == S) {                                                                                             \
    _Pragma("unroll") for (int j = 0; j < FN; ++j) acc[(S >> 2) % FM][j][S & 3] = __builtin_fmaf(vs, bj[j], acc[(S >> 2) % FM][j][S & 3]); \
  }
        SM_COO_CASE(0) SM_COO_CASE(1) SM_COO_CASE(2) SM_COO_CASE(3) SM_COO_CASE(4) SM_COO_CASE(5) SM_COO_CASE(6) SM_COO_CASE(7)
#undef SM_COO_CASE
      }
    }
    if (s + 1 < nst) store_b(cur ^ 1);
    __syncthreads();
  }
  if (bad) wg_bad = 1;
  __syncthreads();
  if (wg_bad) {  // an element of B this tile read does not convert under the call's scale: the tile is not stored
    if (tid == 0) atomicOr(&p.hdr->flag, 1);
    return;
  }
  // ---- epilogue: acc (x 2^-xa x 2^-xb x alpha, one after the other) -> LDS [column][128 rows] -> whole 512-byte column pieces
  const float ia = coo_fast_pow2(-xa), ib = coo_fast_pow2(-xb);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const f4 o = acc[i][j] * ia * ib * p.alpha;
      *reinterpret_cast<f4*>(smem + (16u * j + r16) * CP + (wave * (unsigned)TM + 16u * i + 4u * g) * 4u) = o;
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4096 / NT; ++i) {
    const unsigned q = tid + (unsigned)NT * i, col = q >> 5, ch = q & 31u;
    const int row = m0 + 4 * (int)ch;
    if (n0 + col < p.nv && row < p.m) {  // m % 4 == 0: whole pieces
      const f4 o = *reinterpret_cast<const f4*>(smem + col * CP + ch * 16u);
      __builtin_nontemporal_store(o, reinterpret_cast<f4*>(p.C + (size_t)(n0 + col) * p.m + row));
    }
  }
}

// ---- producer / consumer form (k % 4 == 0): what the kernel above cannot do is keep B's loads more than one stage ahead -- B has to pass
// through registers to become fp16, and with 64 accumulators a wave has none to spare -- so each of its stages waits out one load latency.
// Here a workgroup is EIGHT waves.  Waves 4-7 only bring B in: LDS-DMA (no registers) of the raw fp32 columns into a staging ring three stages
// deep, each lane then reads back the 16 bytes it fetched itself (so the ring needs no synchronisation beyond the wave's own vmcnt), scales,
// checks the range, rounds to fp16 and writes the [column][64 k] image of the NEXT stage.  Waves 0-3 only multiply: A fragments and bucket
// entries from registers loaded two stages ahead, the SMFMACs and the bucket multiply-adds of the CURRENT stage.  One barrier per stage; the
// conversion's vector ALU work runs under the other waves' matrix instructions.  One workgroup per CU (128 KiB of LDS).
__device__ __attribute__((aligned(256))) const unsigned char sm_coo_zero_page[256] = {0};

__global__ __launch_bounds__(512, 2) void spmm_coo_smfmac_pc_kernel(const CooSmArgs p) {
  constexpr int BM = 128, BN = 128, FM = 2, FN = 8, SB = BN * 128, CP = BM * 4 + 16, RS = 3, SSTG = BN * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 images of SB][RS staging slots of SSTG]; epilogue: [BN][CP] over the front
  __shared__ int wg_bad;
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (__builtin_amdgcn_readfirstlane(p.hdr->flag) != 0) return;  // A left the range / a bucket overflowed: C untouched, the caller falls back
  if (tid == 0) wg_bad = 0;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned per_slice = (unsigned)p.tiles_m * (unsigned)p.slice_w;
  const unsigned sl = lid / per_slice, rem = lid - sl * per_slice;
  const unsigned width = (sl + 1u) * (unsigned)p.slice_w <= (unsigned)p.tiles_nv ? (unsigned)p.slice_w : (unsigned)p.tiles_nv - sl * (unsigned)p.slice_w;
  const unsigned tile_m = rem / width, tile_nv = sl * (unsigned)p.slice_w + (rem - tile_m * width);
  const int m0 = (int)tile_m * BM;
  const long long n0 = (long long)tile_nv * BN;
  const int xb = coo_fast_scale_exp(p.hdr->max_b, 12), xa = coo_fast_scale_exp(p.hdr->max_a, 13);
  const int nst = p.nst;
  char* const stg = smem + 2 * SB;
  const unsigned g = lane >> 4, r16 = lane & 15u;
  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  if (wave >= 4u) {
    // ================= producers: wave w brings columns 32 (w - 4) .. + 31, lane -> (column + (lane >> 4), 16-byte piece lane & 15)
    const unsigned pw = wave - 4u, piece = lane & 15u, csub = lane >> 4;
    const float scb = coo_fast_pow2(xb);
    const char* bsrc[8];
    unsigned bdst[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned col = pw * 32u + 4u * i + csub;
      bsrc[i] = n0 + col < p.nv ? reinterpret_cast<const char*>(p.B + (size_t)(n0 + col) * p.k + 4u * piece) : nullptr;
      bdst[i] = a_off(col, piece >> 1) + 8u * (piece & 1u);
    }
    const char* const zero = reinterpret_cast<const char*>(sm_coo_zero_page) + 16u * piece;
    auto dma = [&](int s) {  // always eight instructions (stages past the end and masked lanes read the zero page): vmcnt counts whole stages
      char* dst = stg + (s % RS) * SSTG + pw * 8192u;
      const bool kin = s < nst && s * 64 + 4 * (int)piece < p.k && !SM_COO_ABL(2);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const char* src = kin && bsrc[i] ? bsrc[i] + (size_t)s * 256 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + i * 1024), 16, 0, 0);
      }
    };
    bool bad = false;
    auto convert = [&](int s) {  // the lane's own eight pieces of stage s: staging -> scaled fp16 image
      const char* src = stg + (s % RS) * SSTG + pw * 8192u + 16u * lane;
      char* Bs = smem + (s & 1) * SB;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f4 x = *reinterpret_cast<const f4*>(src + i * 1024) * scb;
        bad |= coo_fast_out_of_range(x[0]) || coo_fast_out_of_range(x[1]) || coo_fast_out_of_range(x[2]) || coo_fast_out_of_range(x[3]);
        typedef _Float16 hv4 __attribute__((ext_vector_type(4)));
        const hv4 h = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
        *reinterpret_cast<hv4*>(Bs + bdst[i]) = h;
      }
    };
    dma(0);
    dma(1);
    dma(2);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // stage 0 has landed
    convert(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int s = 0; s < nst; ++s) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads of staging slot s % RS (stage s, converted last time round) are done
      dma(s + 3);
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // stage s + 1 has landed (s + 2 and s + 3 may still be in flight)
      if (s + 1 < nst) convert(s + 1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the ring in flight when the epilogue image takes the LDS over
    if (bad) wg_bad = 1;
  } else {
    // ================= consumers: wave w multiplies rows 32 w .. + 31 x all 128 columns
    size_t aoff[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int row = m0 + (int)wave * 32 + 16 * i + (int)r16;
      row = row < p.m ? row : p.m - 1;
      aoff[i] = (size_t)row * 4 + g;
    }
    u4 ah[3][FM], al[3][FM];
    unsigned am[3][FM];
    u2 rent[3];
    // The loads are issued by hand and counted by hand: always seven per stage (a stage past the end re-reads the last one), so that
    // "s_waitcnt vmcnt(14)" after issuing stage s + 2 means exactly "stage s has arrived".  Left to the compiler, the loop-carried
    // counters made it drain vmcnt to 0 in front of the first use -- the loads it had just issued included.
    auto load_ar = [&](int s, auto slot_c) {
      constexpr int slot = decltype(slot_c)::value;
      s = s < nst ? s : nst - 1;
      const size_t so = (size_t)s * p.m * 4;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const u4* ph = reinterpret_cast<const u4*>(p.hi) + so + aoff[i];
        const u4* pl = reinterpret_cast<const u4*>(p.lo) + so + aoff[i];
        const unsigned short* pm = p.meta + so + aoff[i];
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ah[slot][i]) : "v"(ph) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(al[slot][i]) : "v"(pl) : "memory");
        asm volatile("global_load_ushort %0, %1, off" : "=v"(am[slot][i]) : "v"(pm) : "memory");
      }
      const u2* pr = p.rlist + (((size_t)tile_m * nst + s) * 4 + wave) * CS_SEG + lane;
      u2& re = rent[slot];
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(re) : "v"(pr) : "memory");
    };
    auto stage = [&](int s, auto slot_c, auto next_c) {
      constexpr int slot = decltype(slot_c)::value;
      load_ar(s + 2, next_c);  // two stages ahead, into the slot stage s - 1 used
      static_assert(FM == 2, "seven loads per stage");
      u2& re = rent[slot];
      asm volatile("s_waitcnt vmcnt(14)"
                   : "+v"(ah[slot][0]), "+v"(ah[slot][1]), "+v"(al[slot][0]), "+v"(al[slot][1]), "+v"(am[slot][0]), "+v"(am[slot][1]), "+v"(re)
                   :: "memory");
      const char* Bs = smem + (s & 1) * SB;
      if (!SM_COO_ABL(4))
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const unsigned col = 16u * j + r16;
        const u4 b0 = *reinterpret_cast<const u4*>(Bs + a_off(col, g));
        const u4 b1 = *reinterpret_cast<const u4*>(Bs + a_off(col, 4u + g));
        typedef uint32_t u8v __attribute__((ext_vector_type(8)));
        const h16 bf = __builtin_bit_cast(h16, u8v{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]});
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          acc[i][j] = smfmac16<false>(__builtin_bit_cast(h8, ah[slot][i]), bf, acc[i][j], (int)am[slot][i]);
          acc[i][j] = smfmac16<false>(__builtin_bit_cast(h8, al[slot][i]), bf, acc[i][j], (int)am[slot][i]);
        }
      }
      {
        const u2 my = rent[slot];
        const int ne = SM_COO_ABL(1) ? 0 : (int)__builtin_popcountll(__ballot(my[0] != CS_EMPTY));  // the valid slots are a prefix of the segment
        for (int e = 0; e < ne; ++e) {
          const unsigned pos = (unsigned)__builtin_amdgcn_readlane((int)my[0], e);
          const int vbits = __builtin_amdgcn_readlane((int)my[1], e);
          const float v = __builtin_bit_cast(float, vbits);
          const unsigned rr = pos & 127u, kk = pos >> 8;
          const float vs = ((rr >> 2) & 3u) == g ? v : 0.f;
          float bj[FN];
#pragma unroll
          for (int j = 0; j < FN; ++j)
            bj[j] = (float)*reinterpret_cast<const _Float16*>(Bs + a_off(16u * j + r16, kk >> 3) + 2u * (kk & 7u));
          const unsigned sel = ((rr >> 4) & 1u) * 4u + (rr & 3u);
#define SM_COO_CASE(S)                                                                                                         \
  if (sel == S) {                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < FN; ++j) acc[S >> 2][j][S & 3] = __builtin_fmaf(vs, bj[j], acc[S >> 2][j][S & 3]); \
  }
          SM_COO_CASE(0) SM_COO_CASE(1) SM_COO_CASE(2) SM_COO_CASE(3) SM_COO_CASE(4) SM_COO_CASE(5) SM_COO_CASE(6) SM_COO_CASE(7)
#undef SM_COO_CASE
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the A / entry loads of later stages stay in flight)
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;
    load_ar(0, c0{});
    load_ar(1, c1{});
    asm volatile("s_barrier" ::: "memory");  // image 0 is in place
    for (int s = 0; s < nst; s += 3) {
      stage(s, c0{}, c2{});
      if (s + 1 < nst) stage(s + 1, c1{}, c0{});
      if (s + 2 < nst) stage(s + 2, c2{}, c1{});
    }
    //  the last two stages issued their clamped look-ahead loads (up to 14, into ah / al / am / rent) and nothing above waits
    // for them: the compiler does not see hand-issued loads, considers those registers dead and may hand them to the epilogue's temporaries,
    // and the workgroup barrier below carries no vmcnt wait.  Drain them here -- the registers are named as read-write operands so that none
    // is reused before the wait -- exactly as the producers do before their epilogue.
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ah[0][0]), "+v"(ah[0][1]), "+v"(ah[1][0]), "+v"(ah[1][1]), "+v"(ah[2][0]), "+v"(ah[2][1]), "+v"(al[0][0]), "+v"(al[0][1]), "+v"(al[1][0]),
                   "+v"(al[1][1]), "+v"(al[2][0]), "+v"(al[2][1])
                 :: "memory");
    asm volatile("" : "+v"(am[0][0]), "+v"(am[0][1]), "+v"(am[1][0]), "+v"(am[1][1]), "+v"(am[2][0]), "+v"(am[2][1]), "+v"(rent[0]), "+v"(rent[1]), "+v"(rent[2]));
  }
  __syncthreads();
  if (wg_bad) {  // an element of B this tile read does not convert under the call's scale: the tile is not stored
    if (tid == 0) atomicOr(&p.hdr->flag, 1);
    return;
  }
  const float ia = coo_fast_pow2(-xa), ib = coo_fast_pow2(-xb);
  if (wave < 4u) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f4 o = acc[i][j] * ia * ib * p.alpha;
        *reinterpret_cast<f4*>(smem + (16u * j + r16) * CP + (wave * 32u + 16u * i + 4u * g) * 4u) = o;
      }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned q = tid + 512u * i, col = q >> 5, ch = q & 31u;
    const int row = m0 + 4 * (int)ch;
    if (n0 + col < p.nv && row < p.m) {  // m % 4 == 0: whole pieces
      const f4 o = *reinterpret_cast<const f4*>(smem + col * CP + ch * 16u);
      __builtin_nontemporal_store(o, reinterpret_cast<f4*>(p.C + (size_t)(n0 + col) * p.m + row));
    }
  }
}

static size_t coo_smfmac_layout(size_t m, size_t k, size_t* o_hi, size_t* o_lo, size_t* o_meta, size_t* o_list) {
  const size_t kc = round_up(k, 64), nst = kc / 64, tiles_m = ceil_div(m, (size_t)128);
  size_t o = COO_FAST_HDR_BYTES;
  o += round_up(m * kc * 4, 256);
  if (o_hi) *o_hi = o;
  o += round_up(nst * m * 64, 256);
  if (o_lo) *o_lo = o;
  o += round_up(nst * m * 64, 256);
  if (o_meta) *o_meta = o;
  o += round_up(nst * m * 8, 256);
  if (o_list) *o_list = o;
  o += tiles_m * nst * CS_CAP * 8;
  return o;
}

size_t coo_smfmac_workspace(size_t m, size_t k, size_t nv) {
  if (m == 0 || k == 0 || nv == 0 || m > 0x7fffff00ull || k > 0x7fffff00ull || m * round_up(k, 64) > ((size_t)1 << 40)) return 0;
  return coo_smfmac_layout(m, k, nullptr, nullptr, nullptr, nullptr);
}

bool coo_smfmac_takes(size_t m, size_t k, size_t nnz, size_t nv, const float* B, const float* C, float beta) {
  if (beta != 0.0f || m % 4 != 0 || m < 4 || k == 0 || nv == 0 || !aligned16(B) || !aligned16(C) || coo_smfmac_workspace(m, k, nv) == 0) return false;
  if (nnz * 5 > m * k) return false;  // denser than 20 %: the buckets grow with the square of the density, the dense-MFMA pipeline takes it
  const size_t tiles = ceil_div(m, (size_t)128) * ceil_div(nv, (size_t)128), nst = ceil_div(k, (size_t)64);
  if (tiles > 0x7fffffffull || nv > 0x7fffffffull) return false;
  // Where both matrix-core forms apply, this one is taken where it measured faster on the same box (profiles/coo_forms_r05ai.txt, coo_forms_r05al.txt; us per
  // call, dense-MFMA pipeline -> this form): k <= 128 (12544 x 64 x 64 59 -> 46, 12544 x 256 x 64 127 -> 104, 3136 x 512 x 128 98 -> 85) and matrices of at
  // most 256 rows, where B is the traffic and the producer / consumer kernel streams it once, converting on the way (196 x 512 x 4608 189 -> 135,
  // 196 x 512 x 2048 98 -> 73, 196 x 2048 x 512 123 -> 86).  Elsewhere a 128 x 128 tile pulls 49 KB per stage through the CU's L1 for 128 SMFMACs -- the
  // L2-to-CU rate, not the matrix pipe, sets the pace (everything but the loads switched off: still 100 of 130 us, coo_ablate_r05al.txt) -- and the pipeline's
  // 256 x 256 tiles on fp16 operands win (3136 x 128 x 1152 106 vs 154, 784 x 256 x 2304 111 vs 148, 12544 x 64 x 576 141 vs 173).
  const bool dense_form_applies = k % 64 == 0 && m >= 8;
  static const int rule_env = tuning_int("SM_COO_SMFMAC", 1);  // tuning aid: 2 = wherever this form can run (the caller handles 0 = never)
  if (rule_env != 2 && dense_form_applies && !(nst <= 2 || m <= 256)) return false;
  return true;
}

int coo_smfmac_product(size_t m, size_t k, size_t nnz, size_t nv, const int* rows, const int* cols, const float* vals, const float* B, float* C, float alpha,
                       void* workspace, hipStream_t st) {
  char* ws = (char*)workspace;
  size_t o_hi, o_lo, o_meta, o_list;
  coo_smfmac_layout(m, k, &o_hi, &o_lo, &o_meta, &o_list);
  CooSmArgs a = {};
  a.hdr = (CooFastHdr*)ws;
  float* A32 = (float*)(ws + COO_FAST_HDR_BYTES);
  a.A32 = A32;
  a.hi = (half_t*)(ws + o_hi); a.lo = (half_t*)(ws + o_lo); a.meta = (unsigned short*)(ws + o_meta);
  a.rlist = (u2*)(ws + o_list);
  a.B = B; a.C = C;
  a.m = (int)m; a.k = (int)k; a.kc = (int)round_up(k, 64); a.nst = a.kc / 64;
  a.tiles_m = (int)ceil_div(m, (size_t)128); a.tiles_nv = (int)ceil_div(nv, (size_t)128);
  a.nv = (long long)nv;
  a.alpha = alpha;
  {  // column tiles per slice: the slice's part of B (fp32) inside ~2.5 MB of an XCD's 4 MiB L2
    static const int slice_env = tuning_int("SM_COO_SLICE", 0);
    const size_t tile_b = (size_t)128 * k * 4;
    size_t w = tile_b ? ((size_t)5 << 19) / tile_b : 1;
    w = w < 1 ? 1 : w;
    if (slice_env > 0) w = (size_t)slice_env;
    a.slice_w = (int)(w > (size_t)a.tiles_nv ? (size_t)a.tiles_nv : w);
  }
  static const int ablate_env = tuning_int("SM_COO_ABLATE", 0);
  a.ablate = ablate_env;
  if (hipMemsetAsync(ws, 0, COO_FAST_HDR_BYTES + m * (size_t)a.kc * 4, st) != hipSuccess) return check_launch("hipMemsetAsync");  // header and the scatter target
  coo_fast_scan(vals, nnz, B, nv * k, a.hdr, st);
  if (nnz) coo_scatter_rows_kernel<<<stream_grid(nnz, 256), 256, 0, st>>>(rows, cols, vals, nnz, m, k, (size_t)a.kc, A32);
  coo_smfmac_compress_kernel<<<(unsigned)(a.tiles_m * a.nst), 256, 0, st>>>(a);
  if (const int rc = check_launch("sm_spmm_coo_f32_fast: 2:4 image of A")) return rc;
  constexpr size_t lds = 128 * (128 * 4 + 16);  // the epilogue image; the two B stages (32 KiB) live inside it
  const unsigned grid = (unsigned)((size_t)a.tiles_m * a.tiles_nv);
  // the producer / consumer kernel (one workgroup per CU) where B is the traffic: few row tiles, several stages; the one-role kernel (two per CU) elsewhere
  static const int pc_env = tuning_int("SM_COO_PC", 1);  // tuning aid: 0 = the one-role kernel everywhere, 2 = this one wherever k % 4 == 0
  if (k % 4 == 0 && pc_env && (pc_env == 2 || (a.tiles_m <= 2 && a.nst >= 3))) {
    constexpr size_t lds_pc = 2 * 128 * 128 + 3 * 128 * 256;  // two images + the staging ring (the epilogue image fits inside)
    static LdsOptIn lds_optin_pc;
    if (const int rc = ensure_dyn_lds(lds_optin_pc, reinterpret_cast<const void*>(&spmm_coo_smfmac_pc_kernel), lds_pc, "spmm_coo_smfmac_pc_kernel")) return rc;
    spmm_coo_smfmac_pc_kernel<<<grid, 512, lds_pc, st>>>(a);
    return check_launch("spmm_coo_smfmac_pc_kernel");
  }
  static const int nw_env = tuning_int("SM_COO_WAVES", 4);  // tuning aid: 8 = eight waves of 16 rows (measured the same to 5 % slower on long K, profiles/coo_waves_r05af.txt)
  static LdsOptIn lds_optin[4];
#define SM_COO_LAUNCH(I, KV, W)                                                                                                                          \
  {                                                                                                                                                      \
    if (const int rc = ensure_dyn_lds(lds_optin[I], reinterpret_cast<const void*>(&spmm_coo_smfmac_kernel<KV, W>), lds, "spmm_coo_smfmac_kernel")) return rc; \
    spmm_coo_smfmac_kernel<KV, W><<<grid, 64 * W, lds, st>>>(a);                                                                                         \
  }
  if (nw_env != 8) {
    if (k % 4 == 0) SM_COO_LAUNCH(0, true, 4) else SM_COO_LAUNCH(1, false, 4)
  } else {
    if (k % 4 == 0) SM_COO_LAUNCH(2, true, 8) else SM_COO_LAUNCH(3, false, 8)
  }
#undef SM_COO_LAUNCH
  return check_launch("spmm_coo_smfmac_kernel");
}

}  // namespace sm
