// spmma_i8.hip -- int8 forms of the 2:4 path (extension; SURVEY.md 8(f) rank 2: the vendor call behind
// include/sparsify.me/spmma.hxx:40-113 lists int8 among its 2:4 types, examples/libcusparse_lt/include/cusparseLt.h:164-169).
//   sm_prune24_i8 (STRIP and TILE) / sm_prune24_check_i8 / sm_compress24_i8 / sm_decompress24_i8: the fp16 rules on |x| of a
//     signed byte (|-128| = 128 > 127; ties keep the lower k index); same blob geometry with 1-byte elements: values
//     [kc/64][M][32 B], metadata [kc/64][M][8 B] (include/sparsifyme.h).
//   sm_spmma_i8: C (int32) = A_2:4 . B (+ C), exact integer arithmetic on v_smfmac_i32_16x16x128_i8.  B is given
//     K-CONTIGUOUS per output column ([n][k], "TN", the layout int8 matrix cores are fed in): a column's 128-k stage
//     piece is one 128-byte row of the LDS image and a lane's operand is two 16-byte chunks of it.
// Operand maps of the instruction, determined on hardware (tools/archive/probe_i8.hip -> profiles/probe_i8_r01.txt):
//   A lane l: row l & 15, g = l >> 4: 16 kept bytes = strips 8 g .. 8 g + 7 of the 128-k stage (dense k 32 g .. + 31),
//     2-bit position code of kept byte e in bits [2 e, 2 e + 1] of the index operand -- i.e. the blob's nibbles of
//     those 8 strips, 4 consecutive metadata bytes, as they are;
//   B lane l: column l & 15, G = l >> 4: bytes 0-15 = dense k 16 G .. 16 G + 15, bytes 16-31 = dense k 64 + 16 G .. + 15;
//   D lane l, register q: row 4 (l >> 4) + q, column l & 15 (the fp16 SMFMAC's map).
// Stage = 128 k = two 64-k planes of the blob: A values image [BM][64 B] (plane 0 | plane 1 per row: lane g's chunk is
// chunk g), metadata [2][BM][8 B], B image [BN][128 B]; all by global_load_lds, ring of 2, one barrier per stage.
#include "select24.h"
#include "mma_tile.h"

namespace sm {

__device__ __attribute__((aligned(256))) const unsigned char sm_zero_page_i8[256] = {0};

__device__ __forceinline__ uint32_t key_i8(uint8_t v) {
  const int x = (int)(int8_t)v;
  return (uint32_t)(x < 0 ? -x : x);  // 0 .. 128
}

// One strip (four signed bytes in a dword) in the composite-key form of select24.h: key_i = |x_i| << 2 | (3 - i) --
// distinct, larger = kept earlier, equal magnitudes ordered by the lower index -- the two largest by a max / min / med
// chain, their low two bits name the kept positions, one v_perm_b32 pulls the two kept bytes out in position order.
//   d = {x3:x2:x1:x0}  ->  kept = {x[p1]:x[p0]} in the low 16 bits,  nib = p0 | p1 << 2  (p0 < p1)
__device__ __forceinline__ void strip_select_i8(uint32_t d, uint32_t& kept, uint32_t& nib) {
  // |x| of the four bytes at once: flip the negative ones and add their sign bit (0x80 -> 0x7f + 1 = 0x80: no carry
  // ever leaves a byte)
  const uint32_t sgn = (d >> 7) & 0x01010101u;
  const uint32_t ab = (d ^ (sgn * 0xffu)) + sgn;
  uint32_t K[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) K[i] = (((ab >> (8 * i)) & 0xffu) << 2) | (uint32_t)(3 - i);
  const uint32_t m01 = K[0] > K[1] ? K[0] : K[1], n01 = K[0] > K[1] ? K[1] : K[0];
  const uint32_t m = m01 > K[2] ? m01 : K[2];
  const uint32_t c01 = m01 < K[2] ? m01 : K[2];
  const uint32_t med = n01 > c01 ? n01 : c01;
  const uint32_t first = m > K[3] ? m : K[3], lo = m > K[3] ? K[3] : m;
  const uint32_t second = lo > med ? lo : med;
  const uint32_t a = first & 3u, b = second & 3u;
  const uint32_t A = a > b ? a : b, B = a > b ? b : a;  // p0 = 3 - A < p1 = 3 - B
  const uint32_t sel = 0x0c0c0000u | ((3u - B) << 8) | (3u - A);
  kept = __builtin_amdgcn_perm(0u, d, sel);
  nib = 15u - (A | (B << 2));
}

// item = 16 dense k of one row (4 strips): one 16-byte load (when aligned) -> 8 kept bytes + 2 metadata bytes
struct I8Item {
  uint8_t e[16];
};
__device__ __forceinline__ void load_item_i8(I8Item& v, const uint8_t* p, size_t nvalid, bool vec) {
  if (vec && nvalid >= 16) {
    *reinterpret_cast<u4*>(v.e) = *reinterpret_cast<const u4*>(p);
  } else {
#pragma unroll
    for (unsigned t = 0; t < 16; ++t) v.e[t] = t < nvalid ? p[t] : (uint8_t)0;
  }
}

__global__ __launch_bounds__(256) void prune_strip_i8_kernel(const uint8_t* A_in, uint8_t* A_out, size_t m, size_t k, size_t ld,
                                                            bool vec) {
  const size_t ipr = (k + 15) / 16, total = m * ipr;
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t row = it / ipr, c = (it - row * ipr) * 16;
    const size_t nvalid = k - c < 16 ? k - c : 16;
    __attribute__((aligned(16))) I8Item v;
    load_item_i8(v, A_in + row * ld + c, nvalid, vec);
#pragma unroll
    for (unsigned s = 0; s < 4; ++s) {
      const unsigned keep = strip_keepmask(key_i8(v.e[4 * s]), key_i8(v.e[4 * s + 1]), key_i8(v.e[4 * s + 2]), key_i8(v.e[4 * s + 3]));
#pragma unroll
      for (unsigned t = 0; t < 4; ++t)
        if (!((keep >> t) & 1u)) v.e[4 * s + t] = 0;
    }
    uint8_t* dst = A_out + row * ld + c;
    if (vec && nvalid >= 16) {
      *reinterpret_cast<u4*>(dst) = *reinterpret_cast<const u4*>(v.e);
    } else {
#pragma unroll
      for (unsigned t = 0; t < 16; ++t)
        if (t < nvalid) dst[t] = v.e[t];
    }
  }
}

// TILE rule (the variant the reference's spmma asks for, spmma.hxx:86): one 4 x 4 tile per thread, magnitudes as fp32
// (exact), the frozen candidate order of select24.h: tile_keepmask.
__global__ __launch_bounds__(256) void prune_tile_i8_kernel(const uint8_t* A_in, uint8_t* A_out, size_t m, size_t k, size_t ld, bool vec) {
  const size_t tpr = (k + 3) / 4, trows = (m + 3) / 4, total = tpr * trows;
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t tr = it / tpr, tc = it - tr * tpr, r0 = tr * 4, c0 = tc * 4;
    const unsigned ncol = k - c0 < 4 ? (unsigned)(k - c0) : 4u;
    uint8_t v[4][4];
    float mag[4][4];
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
      const bool rv = r0 + r < m;
      const uint8_t* p = A_in + (r0 + r) * ld + c0;
      if (rv && vec && ncol == 4) {
        const uint32_t d = *reinterpret_cast<const uint32_t*>(p);
#pragma unroll
        for (unsigned t = 0; t < 4; ++t) v[r][t] = (uint8_t)(d >> (8 * t));
      } else {
#pragma unroll
        for (unsigned t = 0; t < 4; ++t) v[r][t] = (rv && t < ncol) ? p[t] : (uint8_t)0;
      }
#pragma unroll
      for (unsigned t = 0; t < 4; ++t) mag[r][t] = (float)key_i8(v[r][t]);
    }
    const unsigned keep = tile_keepmask(mag);
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
      if (r0 + r >= m) continue;
      uint8_t* p = A_out + (r0 + r) * ld + c0;
#pragma unroll
      for (unsigned t = 0; t < 4; ++t)
        if (t < ncol) p[t] = ((keep >> (4 * r + t)) & 1u) ? v[r][t] : (uint8_t)0;
    }
  }
}

__global__ __launch_bounds__(256) void prune_check_i8_kernel(const uint8_t* A, size_t m, size_t k, size_t ld, bool vec, int* d_valid) {
  const size_t ipr = (k + 15) / 16, total = m * ipr;
  bool bad = false;
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t row = it / ipr, c = (it - row * ipr) * 16;
    __attribute__((aligned(16))) I8Item v;
    load_item_i8(v, A + row * ld + c, k - c < 16 ? k - c : 16, vec);
#pragma unroll
    for (unsigned s = 0; s < 4; ++s) {
      unsigned nnz = 0;
#pragma unroll
      for (unsigned t = 0; t < 4; ++t) nnz += v.e[4 * s + t] != 0;
      bad |= nnz > 2;
    }
  }
  if (__any(bad)) {
    if ((threadIdx.x & 63) == 0) raise_flag(d_valid);
  }
}

// Items (16 dense k of one row -> 8 kept bytes + 2 metadata bytes) are walked so that a row's 128 input bytes of a PAIR
// of planes are read by 8 consecutive lanes (whole cache lines; walking plane by plane reads every line twice, half
// each time: 2.3-2.9 TB/s) -- item it = ((pair * M + R) * 8 + j8): plane 2 pair + j8 / 4, quarter j8 % 4.  The writes
// are then runs of 32 B of values and 8 B of metadata per row and plane, consecutive rows adjacent.
__global__ __launch_bounds__(256) void compress_i8_kernel(const uint8_t* A, size_t m, size_t k, size_t ld, size_t strideA, size_t kc,
                                                         size_t M, uint8_t* vals, unsigned char* meta, bool vec) {
  // blockIdx.y = plane pair, so that no item needs a 64-bit division; contiguous batches (strideA == m * ld) are one
  // tall matrix and need none for the row either
  const size_t nplanes = kc / 64, total = M * 8, sp = blockIdx.y;
  const bool tall = strideA == m * ld;
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t R = it >> 3, j8 = it & 7, s = 2 * sp + (j8 >> 2), c = s * 64 + (j8 & 3) * 16;
    if (s >= nplanes) continue;  // odd plane count: the last pair has one plane
    const size_t o = (s * M + R) * 4 + (j8 & 3);  // output item: 8 value bytes at 8 o, 2 metadata bytes at 2 o
    __attribute__((aligned(8))) uint8_t out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned mb = 0x4444u;  // padding strips: positions (0, 1)
    if (c < k) {
      const uint8_t* src = A + R * ld + c;
      if (!tall) {
        const size_t b = R / m, i = R - b * m;
        src = A + b * strideA + i * ld + c;
      }
      __attribute__((aligned(16))) I8Item v;
      load_item_i8(v, src, k - c < 16 ? k - c : 16, vec);
      mb = 0;
      const u4 d4 = *reinterpret_cast<const u4*>(v.e);
      uint32_t kp[4];
#pragma unroll
      for (unsigned st = 0; st < 4; ++st) {
        uint32_t nib;
        strip_select_i8(d4[st], kp[st], nib);  // a strip at or beyond k is all zeros here: keeps (0, 1), nibble 0x4
        mb |= nib << (4 * st);
      }
      *reinterpret_cast<u2*>(out) = u2{kp[0] | (kp[1] << 16), kp[2] | (kp[3] << 16)};
    }
    *reinterpret_cast<u2*>(vals + o * 8) = *reinterpret_cast<const u2*>(out);
    *reinterpret_cast<unsigned short*>(meta + o * 2) = (unsigned short)mb;
  }
}

__global__ __launch_bounds__(256) void decompress_i8_kernel(const uint8_t* vals, const unsigned char* meta, size_t m, size_t k, size_t ld,
                                                           size_t strideA, size_t kc, size_t M, uint8_t* A) {
  const size_t total = M * (kc / 16);
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t t4 = it >> 2, s = t4 / M, R = t4 - s * M, c = s * 64 + (it & 3) * 16;
    if (c >= k) continue;
    const size_t b = R / m, i = R - b * m;
    const unsigned mb = *reinterpret_cast<const unsigned short*>(meta + it * 2);
    uint8_t* dst = A + b * strideA + i * ld + c;
#pragma unroll
    for (unsigned st = 0; st < 4; ++st) {
      const unsigned nib = (mb >> (4 * st)) & 0xfu, p0 = nib & 3u, p1 = nib >> 2;
#pragma unroll
      for (unsigned t = 0; t < 4; ++t)
        if (c + 4 * st + t < k) dst[4 * st + t] = t == p0 ? vals[it * 8 + 2 * st] : (t == p1 ? vals[it * 8 + 2 * st + 1] : (uint8_t)0);
    }
  }
}

// B of the reference's spmma is row-major k x n (spmma.hxx:40-64); sm_spmma_i8 wants it [n][k].  One-off helper for
// the (small, reused) weight operand: 64 x 64 byte tiles through LDS.
__global__ __launch_bounds__(256) void transpose_i8_kernel(const uint8_t* in, uint8_t* out, size_t rows, size_t cols) {
  __shared__ uint8_t tile[64][65];
  const size_t r0 = (size_t)blockIdx.y * 64, c0 = (size_t)blockIdx.x * 64;
  for (unsigned i = threadIdx.x; i < 64 * 64; i += 256) {
    const unsigned r = i >> 6, c = i & 63u;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? in[(r0 + r) * cols + c0 + c] : (uint8_t)0;
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < 64 * 64; i += 256) {
    const unsigned c = i >> 6, r = i & 63u;
    if (r0 + r < rows && c0 + c < cols) out[(c0 + c) * rows + r0 + r] = tile[r][c];
  }
}

// ---------------------------------------------------------------------------------------------
// matmul
// ---------------------------------------------------------------------------------------------
struct SpmmaI8Args {
  const char* vals;
  const char* meta;
  size_t Mtot;
  const int8_t* Ad;  // fused form: the DENSE A (row-major, lda), selected in the consumer's registers
  size_t sA;         //   its batch stride (elements)
  int lda;
  const int8_t* B;  // [n][k] per batch, ldb = k
  int* C;           // int32 output, or
  int8_t* C8;       // requantised output: sat_int8(rne(scale * acc))
  float scale;
  size_t sB, sC;    // batch strides (elements)
  int m, Mrows, N, K, batch, tiles_m, tiles_n, nplanes;
  int accumulate;
};

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));

// FUSED: prune + compress + matmul in one kernel, the int8 counterpart of spmma_f16_fused_direct_kernel: the stage's A
// image is the DENSE tile [BM][128 B] (128 k), no metadata, and the lane that feeds the matrix instruction selects its
// 8 strips (dense k 32 g .. 32 g + 31 = chunks 2 g, 2 g + 1 of its row) in registers: the same kept bytes and codes as
// sm_compress24_i8 would have stored, so the result is bit-identical to compress + spmma; no blob exists.
template <int BN, int WM, int WN, bool FUSED = false>
__global__ __launch_bounds__(64 * WM * WN) void spmma_i8_kernel(const SpmmaI8Args p) {
  constexpr int BM = 128, NW = WM * WN, TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int SA = FUSED ? BM * 128 : BM * 64, SM_ = FUSED ? 0 : 2 * BM * 8, SB = BN * 128, STAGE = SA + SM_ + SB;
  constexpr int A_N = FUSED ? BM / 8 : BM / 16, M_N = FUSED ? 0 : 2, B_N = BN / 8, W = A_N + M_N + B_N;  // 1 KiB DMA wave-instructions per stage
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
  const size_t row_base = (size_t)b * p.m;
  const int8_t* B = p.B + (size_t)b * p.sB;
  int* C = p.C ? p.C + (size_t)b * p.sC : nullptr;
  int8_t* C8 = p.C8 ? p.C8 + (size_t)b * p.sC : nullptr;
  const int mlast = p.Mrows - 1, nlast = p.N - 1;
  const int nkt = (p.nplanes + 1) / 2;
  const bool odd = (p.nplanes & 1) != 0;  // the last stage then has one plane: its second half meets zeros

  // per slot: source of stage 0, per-stage step, LDS offset, and whether it belongs to the stage's second plane / half
  const char* src[SL];
  size_t step[SL];
  unsigned loff[SL];
  bool second[SL];
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const unsigned t = wave + (unsigned)NW * i;
    src[i] = nullptr; step[i] = 0; loff[i] = 0; second[i] = false;
    if (FUSED && t < (unsigned)A_N) {  // 8 rows x 128 B of the dense A
      const unsigned row = 8u * t + (lane >> 3), cs = (lane & 7u) ^ (row & 7u);
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      src[i] = reinterpret_cast<const char*>(p.Ad + (size_t)b * p.sA + (size_t)gr * p.lda) + 16u * cs;
      step[i] = 128;
      loff[i] = t * 1024u;
      second[i] = cs >= 4u;  // k 64 .. 127 of the stage
    } else if (t < (unsigned)A_N) {  // 16 rows x 64 B of kept values: lane -> row 16 t + lane / 4, LDS chunk lane % 4
      const unsigned row = 16u * t + (lane >> 2), cs = (lane & 3u) ^ a64_swz(row);  // source chunk: plane cs >> 1, half cs & 1
      int gr = m0 + (int)row;
      gr = gr < mlast ? gr : mlast;
      src[i] = p.vals + ((size_t)(cs >> 1) * p.Mtot + row_base + (size_t)gr) * 32 + 16u * (cs & 1u);
      step[i] = 2 * p.Mtot * 32;
      loff[i] = t * 1024u;
      second[i] = (cs >> 1) != 0;  // per lane
    } else if (t < (unsigned)(A_N + M_N)) {  // metadata of plane pl: lane -> rows 2 lane, 2 lane + 1
      const unsigned pl = t - A_N;
      int gr = m0 + 2 * (int)lane;
      gr = gr < mlast ? gr : (mlast & ~1);
      src[i] = p.meta + ((size_t)pl * p.Mtot + row_base + (size_t)gr) * 8;
      step[i] = 2 * p.Mtot * 8;
      loff[i] = SA + pl * (BM * 8);
      second[i] = pl != 0;
    } else if (t < (unsigned)W) {  // B: 8 columns x 128 B (k-contiguous)
      const unsigned j = t - A_N - M_N, col = 8u * j + (lane >> 3), cs = (lane & 7u) ^ (col & 7u);
      int gn = n0 + (int)col;
      gn = gn < nlast ? gn : nlast;
      src[i] = reinterpret_cast<const char*>(B + (size_t)gn * p.K) + 16u * cs;
      step[i] = 128;
      loff[i] = SA + SM_ + j * 1024u;
      second[i] = cs >= 4u;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
    const bool tail = odd && kt == nkt - 1;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const unsigned t = wave + (unsigned)NW * i;  // wave-uniform
      if (t >= (unsigned)W) continue;
      const char* g = src[i] + (size_t)kt * step[i];
      // one-plane tail: the absent plane's values and metadata come from a zero page, and B's k 64 .. 127 (past the end
      // of the column) from the same page -- 0 x anything = 0 in integers
      if (tail && second[i]) g = reinterpret_cast<const char*>(sm_zero_page_i8) + 16u * (lane & 7u);
      __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)(base + loff[i]), 16, 0, 0);
    }
  };

  i4v acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = i4v{0, 0, 0, 0};

  if (nkt > 0) stage(0, 0);
  const unsigned g = lane >> 4, r = lane & 15u;
  for (int kt = 0; kt < nkt; ++kt) {
    wait_dma_and_barrier<0>();  // ring of 2: nothing newer than this stage is in flight
    if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
    const char* As = smem + (kt & 1) * STAGE;
    const char* Ms = As + SA;
    const char* Bs = Ms + SM_;
    i4v af[FM];
    int idx[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const unsigned row = wm * TM + i * 16 + r;
      if constexpr (FUSED) {
        const u4 lo = *reinterpret_cast<const u4*>(As + a_off(row, 2u * g));
        const u4 hi = *reinterpret_cast<const u4*>(As + a_off(row, 2u * g + 1u));
        uint32_t kp[8], nb[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          strip_select_i8(lo[t], kp[t], nb[t]);
          strip_select_i8(hi[t], kp[4 + t], nb[4 + t]);
        }
        af[i] = i4v{(int)(kp[0] | (kp[1] << 16)), (int)(kp[2] | (kp[3] << 16)), (int)(kp[4] | (kp[5] << 16)), (int)(kp[6] | (kp[7] << 16))};
        idx[i] = (int)(nb[0] | (nb[1] << 4) | (nb[2] << 8) | (nb[3] << 12) | (nb[4] << 16) | (nb[5] << 20) | (nb[6] << 24) | (nb[7] << 28));
        continue;
      }
      af[i] = *reinterpret_cast<const i4v*>(As + row * 64u + 16u * (g ^ a64_swz(row)));
      idx[i] = *reinterpret_cast<const int*>(Ms + (g >> 1) * (BM * 8) + row * 8u + 4u * (g & 1u));
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const unsigned col = wn * TN + j * 16 + r;
      const u4 lo = *reinterpret_cast<const u4*>(Bs + a_off(col, g));
      const u4 hi = *reinterpret_cast<const u4*>(Bs + a_off(col, 4u + g));
      const i8v bf = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
      for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(af[i], bf, acc[i][j], idx[i], 0, 0);
    }
  }
  __syncthreads();

  // ---- epilogue: a lane holds 4 consecutive ROWS of one column; transpose through LDS, store 16-byte row pieces
  constexpr int CP = BN * 4 + 16;  // bytes per row of the image
  auto quant = [&](int a) -> int {   // sat_int8(rne(scale * acc)): one fp32 multiply, round to nearest even, clamp
    float f = __builtin_rintf(p.scale * (float)a);
    f = f < -128.0f ? -128.0f : (f > 127.0f ? 127.0f : f);
    return (int)f;
  };
  if (C8) {
    const bool q_vec = (p.N % 16 == 0) && ((reinterpret_cast<uintptr_t>(C8) & 15u) == 0);
    if (q_vec) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const unsigned row = wm * TM + i * 16 + 4u * g, col = wn * TN + j * 16 + r;
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<int*>(smem + (row + q) * CP + col * 4) = acc[i][j][q];
        }
      __syncthreads();
      constexpr int NCH = BM * (BN / 16);
      for (unsigned q = tid; q < (unsigned)NCH; q += 64u * NW) {
        const unsigned row = q / (BN / 16), cn = q % (BN / 16);
        const int gr = m0 + (int)row, gc = n0 + 16 * (int)cn;
        if (gr >= p.Mrows || gc >= p.N) continue;  // N % 16 == 0: a chunk is all in or all out
        u4 o;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          const i4v v = *reinterpret_cast<const i4v*>(smem + row * CP + cn * 64 + w4 * 16);
          o[w4] = (unsigned)(quant(v[0]) & 0xff) | ((unsigned)(quant(v[1]) & 0xff) << 8) | ((unsigned)(quant(v[2]) & 0xff) << 16) |
                  ((unsigned)(quant(v[3]) & 0xff) << 24);
        }
        __builtin_nontemporal_store(o, reinterpret_cast<u4*>(C8 + (size_t)gr * p.N + gc));
      }
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int gc = n0 + (int)(wn * TN + j * 16 + r);
          if (gc >= p.N) continue;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int gr = m0 + (int)(wm * TM + i * 16 + 4u * g) + q;
            if (gr < p.Mrows) C8[(size_t)gr * p.N + gc] = (int8_t)quant(acc[i][j][q]);
          }
        }
    }
    return;
  }
  const bool c_vec = (p.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15u) == 0);
  if (c_vec) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const unsigned row = wm * TM + i * 16 + 4u * g, col = wn * TN + j * 16 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<int*>(smem + (row + q) * CP + col * 4) = acc[i][j][q];
      }
    __syncthreads();
    constexpr int NCH = BM * (BN / 4);
    for (unsigned q = tid; q < (unsigned)NCH; q += 64u * NW) {
      const unsigned row = q / (BN / 4), cn = q % (BN / 4);
      const int gr = m0 + (int)row, gc = n0 + 4 * (int)cn;
      if (gr >= p.Mrows || gc >= p.N) continue;
      i4v v = *reinterpret_cast<const i4v*>(smem + row * CP + cn * 16);
      int* dst = C + (size_t)gr * p.N + gc;
      if (p.accumulate) {
        const i4v old = *reinterpret_cast<const i4v*>(dst);
        v += old;
      }
      __builtin_nontemporal_store(v, reinterpret_cast<i4v*>(dst));
    }
  } else {
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
          int* dst = C + (size_t)gr * p.N + gc;
          *dst = p.accumulate ? *dst + acc[i][j][q] : acc[i][j][q];
        }
      }
  }
}

template <int BN, int WM, int WN, bool FUSED = false>
static int launch_spmma_i8(const SpmmaI8Args& a0, hipStream_t st) {
  SpmmaI8Args a = a0;
  a.tiles_m = (a.Mrows + 127) / 128;
  a.tiles_n = (a.N + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.batch;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_spmma_i8: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  constexpr size_t lds_main = 2 * ((FUSED ? (size_t)128 * 128 : (size_t)128 * 64 + 2 * 128 * 8) + (size_t)BN * 128);
  constexpr size_t lds_epi = (size_t)128 * (BN * 4 + 16);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static LdsOptIn lds_optin;
  if (lds > 64 * 1024) {
    if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmma_i8_kernel<BN, WM, WN, FUSED>), lds, "spmma_i8_kernel")) return rc;
  }
  spmma_i8_kernel<BN, WM, WN, FUSED><<<dim3((unsigned)nwg), dim3(64 * WM * WN), lds, st>>>(a);
  return check_launch("spmma_i8_kernel");
}

}  // namespace sm

using namespace sm;

extern "C" {

int sm_transpose_i8(const void* in, void* out, size_t rows, size_t cols, sm_stream_t s) {
  if (!in || !out || in == out) {
    set_error("sm_transpose_i8: invalid argument (out of place only)");
    return SM_STATUS_INVALID_VALUE;
  }
  if (rows == 0 || cols == 0) return SM_STATUS_SUCCESS;
  if (ceil_div(rows, (size_t)64) > 65535 || ceil_div(cols, (size_t)64) > 0x7fffffffull) {
    set_error("sm_transpose_i8: matrix too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  transpose_i8_kernel<<<dim3((unsigned)ceil_div(cols, (size_t)64), (unsigned)ceil_div(rows, (size_t)64)), 256, 0, (hipStream_t)s>>>(
      (const uint8_t*)in, (uint8_t*)out, rows, cols);
  return check_launch("transpose_i8_kernel");
}

int sm_prune24_i8(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, int alg, sm_stream_t s) {
  if (!A_in || !A_out || ld < k) {
    set_error("sm_prune24_i8: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (alg != SM_PRUNE_STRIP && alg != SM_PRUNE_TILE) {
    set_error("sm_prune24_i8: invalid rule");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || k == 0) return SM_STATUS_SUCCESS;
  if (alg == SM_PRUNE_TILE) {
    const bool vec4 = (reinterpret_cast<uintptr_t>(A_in) & 3u) == 0 && (reinterpret_cast<uintptr_t>(A_out) & 3u) == 0 && ld % 4 == 0;
    prune_tile_i8_kernel<<<stream_grid(ceil_div(m, (size_t)4) * ceil_div(k, (size_t)4), 256), 256, 0, (hipStream_t)s>>>(
        (const uint8_t*)A_in, (uint8_t*)A_out, m, k, ld, vec4);
    return check_launch("prune_tile_i8_kernel");
  }
  const bool vec = aligned16(A_in) && aligned16(A_out) && ld % 16 == 0;
  prune_strip_i8_kernel<<<stream_grid(m * ceil_div(k, (size_t)16), 256), 256, 0, (hipStream_t)s>>>((const uint8_t*)A_in, (uint8_t*)A_out, m, k,
                                                                                                     ld, vec);
  return check_launch("prune_strip_i8_kernel");
}

int sm_prune24_check_i8(const void* A, size_t m, size_t k, size_t ld, int* d_valid, sm_stream_t s) {
  if (!A || !d_valid || ld < k) {
    set_error("sm_prune24_check_i8: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (hipMemsetAsync(d_valid, 0, sizeof(int), (hipStream_t)s) != hipSuccess) return check_launch("hipMemsetAsync");
  if (m == 0 || k == 0) return SM_STATUS_SUCCESS;
  prune_check_i8_kernel<<<stream_grid(m * ceil_div(k, (size_t)16), 256), 256, 0, (hipStream_t)s>>>((const uint8_t*)A, m, k, ld,
                                                                                                     aligned16(A) && ld % 16 == 0, d_valid);
  return check_launch("prune_check_i8_kernel");
}

int sm_compress24_i8(const void* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob, sm_stream_t s) {
  if (!A || !blob || ld < k || !aligned16(blob)) {
    set_error("sm_compress24_i8: invalid argument (blob must be 16-byte aligned)");
    return SM_STATUS_INVALID_VALUE;
  }
  const BlobLayout L = blob_layout(m, k, 1, batch);
  if (L.M == 0 || k == 0) return SM_STATUS_SUCCESS;
  if ((L.kc / 64 + 1) / 2 > 65535) {
    set_error("sm_compress24_i8: k too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  hipStream_t st = (hipStream_t)s;
  const size_t vbytes = L.M * (L.kc / 2), mbytes = L.M * (L.kc / 8);
  if (L.meta_off > vbytes && hipMemsetAsync((char*)blob + vbytes, 0, L.meta_off - vbytes, st) != hipSuccess) return check_launch("hipMemsetAsync");
  if (L.total > L.meta_off + mbytes && hipMemsetAsync((char*)blob + L.meta_off + mbytes, 0, L.total - L.meta_off - mbytes, st) != hipSuccess)
    return check_launch("hipMemsetAsync");
  const bool vec = aligned16(A) && ld % 16 == 0 && strideA % 16 == 0;
  compress_i8_kernel<<<dim3(stream_grid(L.M * 8, 256), (unsigned)((L.kc / 64 + 1) / 2)), 256, 0, st>>>((const uint8_t*)A, m, k, ld, strideA, L.kc, L.M, (uint8_t*)blob,
                                                                          (unsigned char*)blob + L.meta_off, vec);
  return check_launch("compress_i8_kernel");
}

int sm_decompress24_i8(const void* blob, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* A, sm_stream_t s) {
  if (!A || !blob || ld < k) {
    set_error("sm_decompress24_i8: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  const BlobLayout L = blob_layout(m, k, 1, batch);
  if (L.M == 0 || k == 0) return SM_STATUS_SUCCESS;
  decompress_i8_kernel<<<stream_grid(L.M * (L.kc / 16), 256), 256, 0, (hipStream_t)s>>>(
      (const uint8_t*)blob, (const unsigned char*)blob + L.meta_off, m, k, ld, strideA, L.kc, L.M, (uint8_t*)A);
  return check_launch("decompress_i8_kernel");
}

}  // extern "C"

static int spmma_i8_entry(const void* blob, const void* B, int32_t* C, int8_t* C8, float scale, size_t m, size_t n, size_t k, size_t batch,
                          size_t strideB, size_t strideC, int accumulate, sm_stream_t stream) {
  if (!blob || !B || (!C && !C8) || !aligned16(blob)) {
    set_error("sm_spmma_i8: invalid argument (blob must be 16-byte aligned)");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull) {
    set_error("sm_spmma_i8: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  // whole 64-k planes of 16-byte chunks; metadata moves as 16-byte row pairs: even row counts
  if (k % 64 != 0 || m % 2 != 0 || !aligned16(B) || strideB % 16 != 0) {
    set_error("sm_spmma_i8: needs k %% 64 == 0, an even m and a 16-byte aligned B ([n][k], k-contiguous)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  const BlobLayout L = blob_layout(m, k, 1, batch);
  SpmmaI8Args a = {};
  a.vals = (const char*)blob;
  a.meta = (const char*)blob + L.meta_off;
  a.Mtot = L.M;
  a.B = (const int8_t*)B;
  a.C = C;
  a.C8 = C8;
  a.scale = scale;
  a.sB = strideB; a.sC = strideC;
  a.m = (int)m; a.Mrows = (int)m; a.N = (int)n; a.K = (int)k; a.nplanes = (int)(L.kc / 64);
  a.batch = (int)batch; a.accumulate = accumulate != 0;
  if (batch > 1 && strideB == 0 && strideC == m * n) {  // shared B + contiguous C: one tall matrix
    a.Mrows = (int)(m * batch);
    a.batch = 1;
  }
  hipStream_t st = (hipStream_t)stream;
  static const int cfg = tuning_int("SM_SPMMA_I8_CFG", 0);  // tuning aid
  if (cfg == 1) return launch_spmma_i8<64, 4, 1>(a, st);
  if (cfg == 2) return launch_spmma_i8<128, 2, 2>(a, st);
  if (cfg == 3) return launch_spmma_i8<128, 2, 4>(a, st);
  if (cfg == 4) return launch_spmma_i8<128, 4, 4>(a, st);
  // narrow outputs: 128 x 64 tiles over 4 waves (more tiles); otherwise 128 x 128 over 8 (tools/archive/i8_probe.py)
  return n <= 128 ? launch_spmma_i8<64, 4, 1>(a, st) : launch_spmma_i8<128, 2, 4>(a, st);
}

static int spmma_fused_i8_entry(const void* A, const void* B, int32_t* C, int8_t* C8, float scale, size_t m, size_t n, size_t k, size_t lda,
                                size_t batch, size_t strideA, size_t strideB, size_t strideC, int accumulate, sm_stream_t stream) {
  if (!A || !B || (!C && !C8) || lda < k) {
    set_error("sm_spmma_fused_i8: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (m * batch > 0x7fffffffull || n > 0x7fffffffull || k > 0x7fffffffull || lda > 0x7fffffffull) {
    set_error("sm_spmma_fused_i8: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  if (k % 64 != 0 || lda % 16 != 0 || strideA % 16 != 0 || strideB % 16 != 0 || !aligned16(A) || !aligned16(B)) {
    set_error("sm_spmma_fused_i8: needs k %% 64 == 0 and 16-byte aligned rows of A and B (use sm_compress24_i8 + sm_spmma_i8)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  SpmmaI8Args a = {};
  a.Ad = (const int8_t*)A; a.sA = strideA; a.lda = (int)lda;
  a.B = (const int8_t*)B;
  a.C = C; a.C8 = C8; a.scale = scale;
  a.sB = strideB; a.sC = strideC;
  a.m = (int)m; a.Mrows = (int)m; a.N = (int)n; a.K = (int)k; a.nplanes = (int)(k / 64);
  a.batch = (int)batch; a.accumulate = accumulate != 0;
  if (batch > 1 && strideB == 0 && strideA == m * lda && strideC == m * n) {
    a.Mrows = (int)(m * batch);
    a.batch = 1;
  }
  hipStream_t st = (hipStream_t)stream;
  static const int cfg = tuning_int("SM_SPMMA_I8_FUSED_CFG", 0);  // tuning aid
  if (cfg == 1) return launch_spmma_i8<64, 4, 1, true>(a, st);
  if (cfg == 2) return launch_spmma_i8<128, 2, 4, true>(a, st);
  if (cfg == 3) return launch_spmma_i8<128, 4, 2, true>(a, st);
  return n <= 64 ? launch_spmma_i8<64, 4, 1, true>(a, st) : launch_spmma_i8<128, 4, 2, true>(a, st);
}

extern "C" {

int sm_spmma_fused_i8(const void* A, const void* B, int32_t* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                      size_t strideB, size_t strideC, int accumulate, sm_stream_t stream) {
  return spmma_fused_i8_entry(A, B, C, nullptr, 1.0f, m, n, k, lda, batch, strideA, strideB, strideC, accumulate, stream);
}
int sm_spmma_fused_i8_q(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                        size_t strideB, size_t strideC, float scale, sm_stream_t stream) {
  return spmma_fused_i8_entry(A, B, nullptr, (int8_t*)C, scale, m, n, k, lda, batch, strideA, strideB, strideC, 0, stream);
}

int sm_spmma_i8(const void* blob, const void* B, int32_t* C, size_t m, size_t n, size_t k, size_t batch, size_t strideB, size_t strideC,
                int accumulate, sm_stream_t stream) {
  return spmma_i8_entry(blob, B, C, nullptr, 1.0f, m, n, k, batch, strideB, strideC, accumulate, stream);
}
int sm_spmma_i8_q(const void* blob, const void* B, void* C, size_t m, size_t n, size_t k, size_t batch, size_t strideB, size_t strideC,
                  float scale, sm_stream_t stream) {
  return spmma_i8_entry(blob, B, nullptr, (int8_t*)C, scale, m, n, k, batch, strideB, strideC, 0, stream);
}

}  // extern "C"
