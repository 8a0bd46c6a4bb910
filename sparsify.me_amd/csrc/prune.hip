// prune.hip -- HBM-bound streaming kernels of the sparsify.me hot path on gfx950:
//   positional sparsify (reference: include/sparsify.me/sparsify.hxx:24-82),
//   2:4 magnitude prune STRIP / TILE, prune check, compress, decompress
//   (reference call sites: include/sparsify.me/spmma.hxx:86-88, 100-103).
// All kernels move 16 B per lane per access, each wave-instruction covering 1 KiB of contiguous
// memory; selection is pure register work (integer compares on sign-cleared bit patterns), and
// the byte-granular metadata is staged through LDS so it leaves the CU as coalesced dwords.
// Semantics are frozen by oracle/sm_oracle.c; results must match it bit for bit.
#include <stdlib.h>

#include "sm_common.h"
#include "select24.h"

namespace sm {

__device__ __forceinline__ float mag_of(uint16_t v) {
  const uint16_t k = v & 0x7fffu;
  if (k > 0x7c00u) return __builtin_inff();
  return (float)__builtin_bit_cast(_Float16, k);
}
// bfloat16 magnitude (the TILE rule adds magnitudes; the STRIP rule, compress and check only compare bit patterns and
// are shared with fp16 as they are)
__device__ __forceinline__ float mag_of_bf16(uint16_t v) {
  const uint32_t k = v & 0x7fffu;
  if (k > 0x7f80u) return __builtin_inff();
  return __builtin_bit_cast(float, k << 16);
}
__device__ __forceinline__ float mag_of(uint32_t v) {
  const uint32_t k = v & 0x7fffffffu;
  if (k > 0x7f800000u) return __builtin_inff();
  return __builtin_bit_cast(float, k);
}
template <bool BF, typename T>
__device__ __forceinline__ float mag_sel(T v) {
  if constexpr (BF) return mag_of_bf16(v);
  else return mag_of(v);
}

// Two tiles of one thread (4 rows x 8 columns, 16-byte accesses): the rule of select24.h on each.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tile_keepmask2(const f2 (&mag)[4][4], unsigned& keep0, unsigned& keep1) {
  float m0[4][4], m1[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      m0[r][c] = mag[r][c][0];
      m1[r][c] = mag[r][c][1];
    }
  keep0 = tile_keepmask(m0);
  __builtin_amdgcn_sched_barrier(0);  // one tile after the other: interleaved, the two rules' live values double the registers
  keep1 = tile_keepmask(m1);
}

// ---------------------------------------------------------------------------------------------
// (a1) positional sparsify, 2x2 blocks (the only shape the reference instantiates,
//      examples/sparsify.cu:46): fused K1 (mask fill) + K2 (scatter) of SURVEY.md 2.2.
// ---------------------------------------------------------------------------------------------
// zmask: bit o set <=> offset o of every block of 4 consecutive elements is zeroed
// (visit order 0,2,1,3 of sparsify.hxx:53-60).  limit = 4 * (m/2) * (n/2).
template <typename VT /*one element's storage type*/>
__global__ __launch_bounds__(256) void sparsify22_kernel(VT* __restrict__ w, uint64_t* __restrict__ mask,
                                                         size_t total, size_t limit, unsigned zmask) {
  constexpr unsigned E = 16 / sizeof(VT);  // elements per 16-byte vector
  const size_t nchunk = (total + 256 * E - 1) / (256 * E);
  for (size_t chunk = blockIdx.x; chunk < nchunk; chunk += gridDim.x) {
    const size_t e0 = (chunk * 256 + threadIdx.x) * E;  // first element of this lane's vector
    if (e0 < limit) {
      if (e0 + E <= total) {
        u4 v = *reinterpret_cast<const u4*>(w + e0);
        VT* ve = reinterpret_cast<VT*>(&v);
#pragma unroll
        for (unsigned j = 0; j < E; ++j)
          if (e0 + j < limit && ((zmask >> ((e0 + j) & 3)) & 1u)) ve[j] = 0;
        *reinterpret_cast<u4*>(w + e0) = v;
      } else {
        for (unsigned j = 0; j < E && e0 + j < total; ++j)
          if (e0 + j < limit && ((zmask >> ((e0 + j) & 3)) & 1u)) w[e0 + j] = 0;
      }
    }
    // mask: this block's chunk spans 256*E entries = 128*E 16-byte vectors, E/2 per lane,
    // each store instruction covering 256 consecutive vectors.
#pragma unroll
    for (unsigned j = 0; j < E / 2; ++j) {
      const size_t me = chunk * 256 * E + ((size_t)j * 256 + threadIdx.x) * 2;
      if (me >= total) continue;
      const uint64_t a = (me < limit && ((zmask >> (me & 3)) & 1u)) ? 0ull : 1ull;
      if (me + 1 < total) {
        const uint64_t b = (me + 1 < limit && ((zmask >> ((me + 1) & 3)) & 1u)) ? 0ull : 1ull;
        typedef uint64_t ul2 __attribute__((ext_vector_type(2)));
        ul2 mv = {a, b};
        *reinterpret_cast<ul2*>(mask + me) = mv;
      } else {
        mask[me] = a;
      }
    }
  }
}

// Generic restatement for any block shape / unaligned buffers: K1 then K2 as two launches,
// following sparsify.hxx:71 and :43-68 literally (slow path, never used by the drivers).
__global__ void mask_fill_kernel(uint64_t* mask, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    mask[i] = 1;
}
template <typename VT>
__global__ void sparsify_generic_kernel(VT* w, uint64_t* mask, size_t nblk, size_t blk_m, size_t blk_n,
                                        size_t nz) {
  for (size_t blk = blockIdx.x * (size_t)blockDim.x + threadIdx.x; blk < nblk;
       blk += (size_t)gridDim.x * blockDim.x) {
    const size_t g = blk * blk_m * blk_n;
    size_t done = 0;
    for (size_t h = 0; h < blk_m; ++h)
      for (size_t ww = 0; ww < blk_n; ++ww) {
        if (done == nz) break;
        const size_t idx = g + h + ww * blk_n;
        w[idx] = 0;
        mask[idx] = 0;
        ++done;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// element access helpers: 8 consecutive k of one row, vector path when aligned and in range
// ---------------------------------------------------------------------------------------------
template <typename T>
struct Vec8;  // 8 elements of storage type T
template <>
struct Vec8<uint16_t> {
  uint16_t e[8];
  __device__ __forceinline__ void load_vec(const uint16_t* p) { *reinterpret_cast<u4*>(e) = *reinterpret_cast<const u4*>(p); }
  __device__ __forceinline__ void store_vec(uint16_t* p) const { *reinterpret_cast<u4*>(p) = *reinterpret_cast<const u4*>(e); }
};
template <>
struct Vec8<uint32_t> {
  uint32_t e[8];
  __device__ __forceinline__ void load_vec(const uint32_t* p) {
    reinterpret_cast<u4*>(e)[0] = reinterpret_cast<const u4*>(p)[0];
    reinterpret_cast<u4*>(e)[1] = reinterpret_cast<const u4*>(p)[1];
  }
  __device__ __forceinline__ void store_vec(uint32_t* p) const {
    reinterpret_cast<u4*>(p)[0] = reinterpret_cast<const u4*>(e)[0];
    reinterpret_cast<u4*>(p)[1] = reinterpret_cast<const u4*>(e)[1];
  }
};

template <typename T>
__device__ __forceinline__ void load8(Vec8<T>& v, const T* p, size_t nvalid, bool vec_ok) {
  if (vec_ok && nvalid >= 8) {
    v.load_vec(p);
  } else {
#pragma unroll
    for (unsigned t = 0; t < 8; ++t) v.e[t] = t < nvalid ? p[t] : (T)0;
  }
}
template <typename T>
__device__ __forceinline__ void store8(const Vec8<T>& v, T* p, size_t nvalid, bool vec_ok) {
  if (vec_ok && nvalid >= 8) {
    v.store_vec(p);
  } else {
#pragma unroll
    for (unsigned t = 0; t < 8; ++t)
      if (t < nvalid) p[t] = v.e[t];
  }
}

// ---------------------------------------------------------------------------------------------
// (a2) prune STRIP: item = 8 consecutive k of one row (two strips)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void prune_strip_kernel(const T* A_in, T* A_out, size_t m, size_t k,
                                                          size_t ld, bool vec_ok) {
  const size_t ipr = (k + 7) / 8;  // items per row
  const size_t total = m * ipr;
  const bool flat = ld == k && (k & 7) == 0;  // rows are back to back: no 64-bit division per item
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t row = flat ? 0 : it / ipr, c = (it - row * ipr) * 8;
    const size_t nvalid = flat ? 8 : (k - c < 8 ? k - c : 8);
    Vec8<T> v;
    load8(v, A_in + row * ld + c, nvalid, vec_ok);
#pragma unroll
    for (unsigned s = 0; s < 2; ++s) {
      const unsigned keep = strip_keepmask(key_of(v.e[4 * s]), key_of(v.e[4 * s + 1]), key_of(v.e[4 * s + 2]),
                                           key_of(v.e[4 * s + 3]));
#pragma unroll
      for (unsigned t = 0; t < 4; ++t)
        if (!((keep >> t) & 1u)) v.e[4 * s + t] = 0;
    }
    store8(v, A_out + row * ld + c, nvalid, vec_ok);
  }
}

// ---------------------------------------------------------------------------------------------
// (a2) prune TILE: item = one 4x4 tile (K3 of SURVEY.md 2.2, the variant spmma.hxx:86 asks for)
// ---------------------------------------------------------------------------------------------
template <typename T, bool BF = false>
__global__ __launch_bounds__(256) void prune_tile_kernel(const T* A_in, T* A_out, size_t m, size_t k,
                                                         size_t ld, bool vec_ok) {
  const size_t tpr = (k + 3) / 4, trows = (m + 3) / 4;
  const size_t total = tpr * trows;
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t tr = it / tpr, tc = it - tr * tpr;
    const size_t r0 = tr * 4, c0 = tc * 4;
    const unsigned ncol = k - c0 < 4 ? (unsigned)(k - c0) : 4u;
    T v[4][4];
    float mag[4][4];
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
      const bool rv = r0 + r < m;
      const T* p = A_in + (r0 + r) * ld + c0;
      if (rv && vec_ok && ncol == 4) {
        if constexpr (sizeof(T) == 2) {
          *reinterpret_cast<u2*>(v[r]) = *reinterpret_cast<const u2*>(p);
        } else {
          *reinterpret_cast<u4*>(v[r]) = *reinterpret_cast<const u4*>(p);
        }
      } else {
#pragma unroll
        for (unsigned t = 0; t < 4; ++t) v[r][t] = (rv && t < ncol) ? p[t] : (T)0;
      }
#pragma unroll
      for (unsigned t = 0; t < 4; ++t) mag[r][t] = mag_sel<BF>(v[r][t]);
    }
    const unsigned keep = tile_keepmask(mag);
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
      if (r0 + r >= m) continue;
#pragma unroll
      for (unsigned t = 0; t < 4; ++t)
        if (!((keep >> (4 * r + t)) & 1u)) v[r][t] = 0;
      T* p = A_out + (r0 + r) * ld + c0;
      if (vec_ok && ncol == 4) {
        if constexpr (sizeof(T) == 2) {
          *reinterpret_cast<u2*>(p) = *reinterpret_cast<const u2*>(v[r]);
        } else {
          *reinterpret_cast<u4*>(p) = *reinterpret_cast<const u4*>(v[r]);
        }
      } else {
#pragma unroll
        for (unsigned t = 0; t < 4; ++t)
          if (t < ncol) p[t] = v[r][t];
      }
    }
  }
}

// Fast path of TILE: 4 rows x 8 columns (two tiles) per thread, 16-byte (fp16) / 2 x 16-byte (fp32) accesses, the
// adds of both tiles packed (tile_keepmask2).  Needs k % 8 == 0 and aligned rows; row tails (m % 4) are zero-filled
// on load and skipped on store.
template <typename T, bool BF = false>
__global__ __launch_bounds__(256) void prune_tile2_kernel(const T* __restrict__ A_in, T* __restrict__ A_out, size_t m,
                                                          size_t k, size_t ld) {
  const size_t ppr = k / 8, trows = (m + 3) / 4;
  const size_t total = ppr * trows;
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t tr = it / ppr, pc = it - tr * ppr;
    const size_t r0 = tr * 4, c0 = pc * 8;
    Vec8<T> v[4];
    f2 mag[4][4];
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
      if (r0 + r < m) {
        v[r].load_vec(A_in + (r0 + r) * ld + c0);
      } else {
#pragma unroll
        for (unsigned t = 0; t < 8; ++t) v[r].e[t] = 0;
      }
#pragma unroll
      for (unsigned t = 0; t < 4; ++t) mag[r][t] = f2{mag_sel<BF>(v[r].e[t]), mag_sel<BF>(v[r].e[4 + t])};
    }
    unsigned keep0, keep1;
    tile_keepmask2(mag, keep0, keep1);
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
      if (r0 + r >= m) continue;
#pragma unroll
      for (unsigned t = 0; t < 4; ++t) {
        if (!((keep0 >> (4 * r + t)) & 1u)) v[r].e[t] = 0;
        if (!((keep1 >> (4 * r + t)) & 1u)) v[r].e[4 + t] = 0;
      }
      v[r].store_vec(A_out + (r0 + r) * ld + c0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// TILE on rows that are NOT whole 16-byte pieces (round 6: k % 8 != 0 or an odd row pitch -- the 7 x 7 x 3 stem layer of every ResNet,
// k = 147: rows of 294 bytes), 16-bit types, ONE contiguous matrix (ld == k): SPAN form.  prune_tile_kernel above then moves single
// halves (16 two-byte loads and stores per tile: 2.8 TB/s of traffic where the aligned kernel reaches 4.6).  Here a workgroup owns
// SPAN_ROWS (a multiple of 8, chosen by the launcher) consecutive rows = ONE contiguous span of 16-byte pieces whatever k is: it comes into LDS by
// plain 16-byte loads (whole lines), the tiles are evaluated out of LDS with two-byte reads (SPAN_ROWS / 4 tile rows x ceil(k / 4) tiles per span,
// row tails and the matrix's last rows completed with virtual zeros as the oracle does), the pruned halves are written back INTO the LDS
// image and the span leaves by 16-byte stores.  In place or out of place (a workgroup owns its rows).  d_valid (may be null): raised when a
// stored strip holds more than two values != 0 -- derived from what is stored, as the one-pass kernels derive it -- so the API sequence
// needs no separate check pass on this shape.  The last span may be short and may end off a 16-byte boundary: its tail bytes move as halves.
// ---------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256) void prune_tile_span_kernel(const uint16_t* A_in, uint16_t* A_out, size_t m, unsigned k,
                                                              int* d_valid, const unsigned SPAN_ROWS /* a multiple of 8: the span is whole 16-byte pieces */) {
  extern __shared__ __attribute__((aligned(16))) uint16_t tspan[];
  const unsigned tid = threadIdx.x;
  const size_t r0 = (size_t)blockIdx.x * SPAN_ROWS;
  const unsigned rows = m - r0 < SPAN_ROWS ? (unsigned)(m - r0) : SPAN_ROWS;
  const size_t e0 = r0 * k;                       // first element of the span: r0 * k * 2 bytes = a multiple of 64
  const unsigned elems = rows * k, pieces = elems / 8u, tail0 = pieces * 8u;
  const u4* src = reinterpret_cast<const u4*>(A_in + e0);
  for (unsigned i = tid; i < pieces; i += 256u) reinterpret_cast<u4*>(tspan)[i] = __builtin_nontemporal_load(src + i);
  for (unsigned i = tail0 + tid; i < elems; i += 256u) tspan[i] = A_in[e0 + i];
  __syncthreads();
  const unsigned tpr = (k + 3u) / 4u, trows = (rows + 3u) / 4u, tiles = tpr * trows;
  bool bad = false;
  for (unsigned t = tid; t < tiles; t += 256u) {
    const unsigned tr = t / tpr, tc = t - tr * tpr, c0 = 4u * tc;
    const unsigned ncol = k - c0 < 4u ? k - c0 : 4u, nrow = rows - 4u * tr < 4u ? rows - 4u * tr : 4u;
    uint16_t v[4][4];
    float mag[4][4];
#pragma unroll
    for (unsigned r = 0; r < 4; ++r)
#pragma unroll
      for (unsigned c = 0; c < 4; ++c) {
        v[r][c] = (r < nrow && c < ncol) ? tspan[(4u * tr + r) * k + c0 + c] : (uint16_t)0;
        mag[r][c] = mag_sel<BF>(v[r][c]);
      }
    const unsigned keep = tile_keepmask(mag);
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
      unsigned nz = 0;
#pragma unroll
      for (unsigned c = 0; c < 4; ++c) {
        const bool kept = (keep >> (4u * r + c)) & 1u;
        if (r < nrow && c < ncol && !kept) tspan[(4u * tr + r) * k + c0 + c] = 0;
        nz += (kept && (v[r][c] & 0x7fffu) != 0) ? 1u : 0u;
      }
      bad |= nz > 2u;
    }
  }
  __syncthreads();
  u4* dst = reinterpret_cast<u4*>(A_out + e0);
  for (unsigned i = tid; i < pieces; i += 256u) dst[i] = reinterpret_cast<const u4*>(tspan)[i];
  for (unsigned i = tail0 + tid; i < elems; i += 256u) A_out[e0 + i] = tspan[i];
  if (d_valid && __any(bad)) {
    if ((tid & 63u) == 0) raise_flag(d_valid);
  }
}

// the span form's launcher: SM_STATUS_NOT_SUPPORTED when the shape is not its (then the element-wise kernel runs)
int prune24_tile_span_u16(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, bool bf, int* d_valid, hipStream_t st) {
  if (ld != k || k == 0 || m == 0 || (size_t)8 * k * 2 > 64 * 1024 || !aligned16(A_in) || !aligned16(A_out) || m * k > ((size_t)1 << 40)) return SM_STATUS_NOT_SUPPORTED;
  // rows per span: the multiple of 8 (<= 64, span within 64 KiB) whose tile count fills whole rounds of the workgroup's 256 threads best
  // (k = 147: 37 tiles per tile row -- 32 rows = 296 tiles = two rounds at 58 %; 24 rows = 222 tiles = one round at 87 %)
  const size_t tpr = (k + 3) / 4;
  unsigned span_rows = 8;
  double best = 0.0;
  for (unsigned r = 8; r <= 64 && (size_t)r * k * 2 <= 64 * 1024; r += 8) {
    const size_t tiles = (size_t)(r / 4) * tpr;
    const double util = (double)tiles / (double)(ceil_div(tiles, (size_t)256) * 256);
    if (util > best + 1e-9) { best = util; span_rows = r; }
  }
  const size_t spans = ceil_div(m, (size_t)span_rows), lds = (size_t)span_rows * k * 2;
  if (spans > 0x7fffffffull) return SM_STATUS_NOT_SUPPORTED;
  if (bf) prune_tile_span_kernel<true><<<(unsigned)spans, 256, lds, st>>>((const uint16_t*)A_in, (uint16_t*)A_out, m, (unsigned)k, d_valid, span_rows);
  else prune_tile_span_kernel<false><<<(unsigned)spans, 256, lds, st>>>((const uint16_t*)A_in, (uint16_t*)A_out, m, (unsigned)k, d_valid, span_rows);
  return check_launch("prune_tile_span_kernel");
}

// ---------------------------------------------------------------------------------------------
// (a2) prune check (K4): any strip with more than two non-zeros -> *d_valid |= 1
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void prune_check_kernel(const T* A, size_t m, size_t k, size_t ld,
                                                          bool vec_ok, int* d_valid) {
  const size_t ipr = (k + 7) / 8;
  const size_t total = m * ipr;
  const bool flat = ld == k && (k & 7) == 0;
  bool bad = false;
  // (four independent loads in flight per lane were measured: 95 vs 79 us on a 462 MB operand, profiles/prune_rates_r04al.txt: slower)
  for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < total; it += (size_t)gridDim.x * 256) {
    const size_t row = flat ? 0 : it / ipr, c = (it - row * ipr) * 8;
    const size_t nvalid = flat ? 8 : (k - c < 8 ? k - c : 8);
    Vec8<T> v;
    load8(v, A + row * ld + c, nvalid, vec_ok);
#pragma unroll
    for (unsigned s = 0; s < 2; ++s) {
      unsigned nnz = 0;
#pragma unroll
      for (unsigned t = 0; t < 4; ++t) nnz += key_of(v.e[4 * s + t]) != 0;
      bad |= nnz > 2;
    }
  }
  if (__any(bad)) {
    if ((threadIdx.x & 63) == 0) raise_flag(d_valid);
  }
}

// ---------------------------------------------------------------------------------------------
// (a3) compress (K5, fused with the STRIP selection).  item = 8 dense k of one blob row -> 4 kept values
//      (one 8- or 16-byte store) + 1 metadata byte.  Both blob sections are stage-major, so item
//      it' = ((s * M + R) * 8 + j8)  (s = 64-k stage, R = blob row, j8 = item of the stage) owns values
//      [4 it', 4 it' + 4) and metadata byte it': the OUTPUT is linear in it'.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void select_item(const Vec8<T>& v, unsigned valid_strips /*bit s: strip s in range*/, T out[4],
                                            unsigned& mb) {
  unsigned nib[2];
#pragma unroll
  for (unsigned s = 0; s < 2; ++s) {
    // a strip wholly at or beyond k keeps the padding nibble 0x4 and zero values
    const unsigned keep = ((valid_strips >> s) & 1u)
                              ? strip_keepmask(key_of(v.e[4 * s]), key_of(v.e[4 * s + 1]), key_of(v.e[4 * s + 2]),
                                               key_of(v.e[4 * s + 3]))
                              : 3u;
    nib[s] = nibble_of(keep);
    // select the two kept values without dynamic register indexing
    const unsigned p0 = nib[s] & 3u, p1 = nib[s] >> 2;
    T a0 = v.e[4 * s], a1 = v.e[4 * s + 1];
    a0 = p0 == 1 ? v.e[4 * s + 1] : a0;
    a0 = p0 == 2 ? v.e[4 * s + 2] : a0;
    a1 = p1 == 2 ? v.e[4 * s + 2] : a1;
    a1 = p1 == 3 ? v.e[4 * s + 3] : a1;
    out[2 * s] = a0;
    out[2 * s + 1] = a1;
  }
  mb = nib[0] | (nib[1] << 4);
}

template <typename T>
__device__ __forceinline__ void store_item(T* vals, size_t itp, const T out[4]) {
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<u2*>(vals + itp * 4) = *reinterpret_cast<const u2*>(out);
  } else {
    *reinterpret_cast<u4*>(vals + itp * 4) = *reinterpret_cast<const u4*>(out);
  }
}

// General form (any k, ld, batch stride, alignment): walks it' linearly, one division by M per item.
template <typename T>
__global__ __launch_bounds__(256) void compress_kernel(const T* A, size_t m, size_t k, size_t ld,
                                                       size_t strideA, size_t kc, size_t M, T* vals,
                                                       unsigned char* meta, bool vec_ok) {
  const size_t total = M * (kc / 8);  // a multiple of 8
  for (size_t itp = blockIdx.x * (size_t)256 + threadIdx.x; itp < total; itp += (size_t)gridDim.x * 256) {
    const size_t t = itp >> 3, s = t / M, R = t - s * M, c = (s * 8 + (itp & 7)) * 8;
    T out[4] = {0, 0, 0, 0};
    unsigned mb = 0x44;
    if (c < k) {
      const size_t b = R / m, i = R - b * m;
      const size_t nvalid = k - c < 8 ? k - c : 8;
      Vec8<T> v;
      load8(v, A + b * strideA + i * ld + c, nvalid, vec_ok);
      select_item(v, 1u | (c + 4 < k ? 2u : 0u), out, mb);
    }
    store_item(vals, itp, out);
    meta[itp] = (unsigned char)mb;
  }
}

// Fast path of compress: dense input with no padding columns (k % 64 == 0), rows and batches back to back,
// 16-byte aligned (every ResNet layer but k = 147).  No LDS, no barrier.  The work is cut into units of
// 8 rows x 1 stage, numbered with the stage running fastest (unit U = row group * nstage + stage); a block
// takes 4 * NLD consecutive units, unit NLD * wave + j being the j-th of the NLD 16-byte loads each lane issues up
// front (A is read exactly once: non-temporal); lane l of a unit holds item j8 = l & 7 of row (l >> 3).
// Reads are therefore runs of up to 2 KiB per row, writes 512 contiguous bytes of values + 64 contiguous
// metadata bytes per unit; the four metadata bytes of a quad's lanes are gathered with DPP quad permutes so
// its first lane writes one dword.  One 32-bit division per unit (wave-uniform).
template <typename T, bool NT, int NLD>
__global__ __launch_bounds__(256) void compress_flat_kernel(const T* __restrict__ A, size_t M, size_t k, unsigned nstage,
                                                            unsigned nunits, T* __restrict__ vals,
                                                            unsigned char* __restrict__ meta) {
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  Vec8<T> v[NLD];
  size_t itp[NLD];
  bool ok[NLD];
#pragma unroll
  for (unsigned j = 0; j < (unsigned)NLD; ++j) {
    const unsigned U = blockIdx.x * (4u * NLD) + (unsigned)NLD * wave + j, rg = U / nstage, s = U - rg * nstage;
    const size_t R = (size_t)rg * 8u + (lane >> 3);
    ok[j] = U < nunits && R < M;
    itp[j] = (((size_t)s * M + R) << 3) + (lane & 7u);
    if (ok[j]) {
      const u4* p = reinterpret_cast<const u4*>(A + R * k + (size_t)s * 64 + (lane & 7u) * 8);
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<u4*>(v[j].e) = NT ? __builtin_nontemporal_load(p) : *p;
      } else {
        reinterpret_cast<u4*>(v[j].e)[0] = NT ? __builtin_nontemporal_load(p) : p[0];
        reinterpret_cast<u4*>(v[j].e)[1] = NT ? __builtin_nontemporal_load(p + 1) : p[1];
      }
    } else {
#pragma unroll
      for (unsigned t = 0; t < 8; ++t) v[j].e[t] = 0;
    }
  }
#pragma unroll
  for (unsigned j = 0; j < (unsigned)NLD; ++j) {
    T out[4];
    unsigned mbu;
    if constexpr (sizeof(T) == 2) {  // composite-key selection, two strips of packed halves (select24.h)
      const u4 d = *reinterpret_cast<const u4*>(v[j].e);
      uint32_t k0, k1, n0, n1;
      strip_select_f16(d[0], d[1], k0, n0);
      strip_select_f16(d[2], d[3], k1, n1);
      *reinterpret_cast<u2*>(out) = u2{k0, k1};
      mbu = n0 | (n1 << 4);
    } else {
      select_item(v[j], 3u, out, mbu);
    }
    const int mb = (int)mbu;
    // bytes of the quad's four lanes -> one dword (same value in all four lanes)
    const unsigned b0 = (unsigned)__builtin_amdgcn_update_dpp(0, mb, 0x00, 0xf, 0xf, true);
    const unsigned b1 = (unsigned)__builtin_amdgcn_update_dpp(0, mb, 0x55, 0xf, 0xf, true);
    const unsigned b2 = (unsigned)__builtin_amdgcn_update_dpp(0, mb, 0xaa, 0xf, 0xf, true);
    const unsigned b3 = (unsigned)__builtin_amdgcn_update_dpp(0, mb, 0xff, 0xf, 0xf, true);
    if (ok[j]) {  // a quad shares its row and stage: all in or all out
      store_item(vals, itp[j], out);
      if ((lane & 3u) == 0) *reinterpret_cast<unsigned*>(meta + itp[j]) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
  }
}

// fp16 rows whose byte length is not a multiple of 16 (k = 147: the 7 x 7 x 3 first layer), rows and batches back
// to back: a block takes 32 rows; their bytes -- one contiguous span of the input -- are copied to LDS with aligned
// 16-byte loads (coalesced whatever the row pitch), then thread (row = t / 8, j8 = t % 8) walks the stages, reads its
// 8 halves with 2-byte LDS reads (zero beyond k: the padding semantics then fall out of the selection itself, a
// strip of zeros keeps positions 0,1 = nibble 0x4) and stores like the fast path.  80 -> 46 us on 12544 x 147, b = 32.
__global__ __launch_bounds__(256) void compress_rowspan_f16_kernel(const uint16_t* __restrict__ A, size_t M, size_t k,
                                                                   size_t ld, unsigned nstage,
                                                                   uint16_t* __restrict__ vals,
                                                                   unsigned char* __restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char span[];
  const size_t R0 = (size_t)blockIdx.x * 32;
  const unsigned rows = (unsigned)(M - R0 < 32 ? M - R0 : 32);
  const size_t total = M * ld * 2;                 // bytes of the whole input
  const size_t b0 = R0 * ld * 2, b1 = (R0 + rows) * ld * 2;
  const size_t a0 = b0 & ~(size_t)15;              // A is 16-byte aligned (launcher)
  const unsigned char* src = reinterpret_cast<const unsigned char*>(A);
  for (size_t off = a0 + (size_t)threadIdx.x * 16; off < b1; off += 256 * 16) {
    if (off + 16 <= total) {
      *reinterpret_cast<u4*>(span + (off - a0)) = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + off));
    } else {  // the last, partial chunk of the buffer: never read past its end
      for (size_t i = off; i < total; i += 2)
        *reinterpret_cast<uint16_t*>(span + (i - a0)) = *reinterpret_cast<const uint16_t*>(src + i);
    }
  }
  __syncthreads();
  const unsigned lane = threadIdx.x & 63u, row = threadIdx.x >> 3, j8 = threadIdx.x & 7u;
  const bool ok = row < rows;
  const unsigned char* rp = span + (b0 - a0) + (size_t)(ok ? row : 0) * ld * 2;
  for (unsigned s = 0; s < nstage; ++s) {
    const size_t c = ((size_t)s * 8 + j8) * 8;
    uint16_t e[8];
#pragma unroll
    for (unsigned t = 0; t < 8; ++t) e[t] = (ok && c + t < k) ? *reinterpret_cast<const uint16_t*>(rp + (c + t) * 2) : (uint16_t)0;
    uint32_t k0, k1, n0, n1;
    strip_select_f16((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16), k0, n0);
    strip_select_f16((uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16), k1, n1);
    const int mb = (int)(n0 | (n1 << 4));
    const unsigned q0 = (unsigned)__builtin_amdgcn_update_dpp(0, mb, 0x00, 0xf, 0xf, true);
    const unsigned q1 = (unsigned)__builtin_amdgcn_update_dpp(0, mb, 0x55, 0xf, 0xf, true);
    const unsigned q2 = (unsigned)__builtin_amdgcn_update_dpp(0, mb, 0xaa, 0xf, 0xf, true);
    const unsigned q3 = (unsigned)__builtin_amdgcn_update_dpp(0, mb, 0xff, 0xf, 0xf, true);
    if (ok) {
      const size_t itp = (((size_t)s * M + R0 + row) << 3) + j8;
      *reinterpret_cast<u2*>(vals + itp * 4) = u2{k0, k1};
      if ((lane & 3u) == 0) *reinterpret_cast<unsigned*>(meta + itp) = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void decompress_kernel(const T* vals, const unsigned char* meta, size_t m,
                                                         size_t k, size_t ld, size_t strideA, size_t kc,
                                                         size_t M, T* A, bool vec_ok) {
  const size_t total = M * (kc / 8);
  for (size_t itp = blockIdx.x * (size_t)256 + threadIdx.x; itp < total; itp += (size_t)gridDim.x * 256) {
    const size_t t = itp >> 3, s = t / M, R = t - s * M, c = (s * 8 + (itp & 7)) * 8;
    if (c >= k) continue;
    const size_t b = R / m, i = R - b * m;
    const unsigned mb = meta[itp];
    T in[4];
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<u2*>(in) = *reinterpret_cast<const u2*>(vals + itp * 4);
    } else {
      *reinterpret_cast<u4*>(in) = *reinterpret_cast<const u4*>(vals + itp * 4);
    }
    Vec8<T> v;
#pragma unroll
    for (unsigned s2 = 0; s2 < 2; ++s2) {
      const unsigned nib = (mb >> (4 * s2)) & 0xfu, p0 = nib & 3u, p1 = nib >> 2;
#pragma unroll
      for (unsigned t2 = 0; t2 < 4; ++t2) v.e[4 * s2 + t2] = t2 == p0 ? in[2 * s2] : (t2 == p1 ? in[2 * s2 + 1] : (T)0);
    }
    const size_t nvalid = k - c < 8 ? k - c : 8;
    store8(v, A + b * strideA + i * ld + c, nvalid, vec_ok);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <typename VT>
static int launch_sparsify(void* weights, uint64_t* mask, size_t m, size_t n, size_t blk_m, size_t blk_n,
                           float sf, hipStream_t st) {
  const size_t total = m * n;
  const size_t blk_size = blk_m * blk_n;
  const size_t nblk = (m / blk_m) * (n / blk_n);
  const float nzf = floorf((float)blk_size * sf);  // sparsify.hxx:42
  size_t nz = nzf <= 0.0f ? 0 : (size_t)nzf;
  if (nz > blk_size) nz = blk_size;
  if (total == 0) return SM_STATUS_SUCCESS;
  if (blk_m == 2 && blk_n == 2 && aligned16(weights) && aligned16(mask)) {
    static const unsigned ZM[5] = {0x0u, 0x1u, 0x5u, 0x7u, 0xfu};  // visit order 0,2,1,3
    constexpr unsigned E = 16 / sizeof(VT);
    const unsigned grid = stream_grid(ceil_div(total, E), 256);
    sparsify22_kernel<VT><<<grid, 256, 0, st>>>((VT*)weights, mask, total, nblk * 4, ZM[nz]);
    return check_launch("sparsify22_kernel");
  }
  // generic: refuse what would write outside the buffer (the reference would, sparsify.hxx:60)
  if (nblk > 0 && nz > 0) {
    size_t max_off = 0, cnt = 0;
    for (size_t h = 0; h < blk_m && cnt < nz; ++h)
      for (size_t w = 0; w < blk_n && cnt < nz; ++w, ++cnt)
        if (h + w * blk_n > max_off) max_off = h + w * blk_n;
    if ((nblk - 1) * blk_size + max_off >= total) {
      set_error("sm_sparsify_positional: block shape %zux%zu indexes outside the %zux%zu buffer", blk_m, blk_n, m, n);
      return SM_STATUS_INVALID_VALUE;
    }
  }
  mask_fill_kernel<<<stream_grid(total, 256), 256, 0, st>>>(mask, total);
  if (nblk > 0 && nz > 0)
    sparsify_generic_kernel<VT><<<stream_grid(nblk, 256), 256, 0, st>>>((VT*)weights, mask, nblk, blk_m, blk_n, nz);
  return check_launch("sparsify_generic_kernel");
}

template <typename T>
static bool vec_ok_2d(const void* a, const void* b, size_t ld, size_t stride) {
  constexpr size_t per16 = 16 / sizeof(T);
  // 8 elements are moved as one (f16) or two (f32) 16-byte accesses at element offsets that are
  // multiples of 8, so rows and batches must start on 16-byte boundaries.
  return aligned16(a) && aligned16(b) && ld % per16 == 0 && stride % per16 == 0;
}

template <typename T, bool BF = false>
static int launch_prune(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, int alg, hipStream_t st) {
  if (!A_in || !A_out || ld < k || (alg != SM_PRUNE_TILE && alg != SM_PRUNE_STRIP)) {
    set_error("sm_prune24: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (m == 0 || k == 0) return SM_STATUS_SUCCESS;
  if (alg == SM_PRUNE_STRIP) {
    const bool vec_ok = vec_ok_2d<T>(A_in, A_out, ld, 0);
    prune_strip_kernel<T><<<stream_grid(m * ceil_div(k, 8), 256, true), 256, 0, st>>>((const T*)A_in, (T*)A_out, m, k, ld, vec_ok);
    return check_launch("prune_strip_kernel");
  }
  // TILE moves 4 elements per row access: 8-byte (f16) / 16-byte (f32) alignment
  const size_t per = 4;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(A_in) % (per * sizeof(T)) == 0) &&
                      (reinterpret_cast<uintptr_t>(A_out) % (per * sizeof(T)) == 0) && ld % per == 0;
  // fp16 is bound by the rule's arithmetic (two tiles per thread with packed adds: 3.76 -> 3.20 ms on the ResNet-50
  // table); fp32 moves twice the bytes for the same arithmetic and stays on the one-tile kernel (5.1 TB/s)
  if (sizeof(T) == 2 && k % 8 == 0 && vec_ok_2d<T>(A_in, A_out, ld, 0)) {
    prune_tile2_kernel<T, BF><<<stream_grid(ceil_div(m, 4) * (k / 8), 256), 256, 0, st>>>((const T*)A_in, (T*)A_out, m, k, ld);
    return check_launch("prune_tile2_kernel");
  }
  if (sizeof(T) == 2) {  // (round 6) ragged rows of one contiguous 16-bit matrix: the span form (no flag wanted here)
    const int rc = prune24_tile_span_u16(A_in, A_out, m, k, ld, BF, nullptr, st);
    if (rc != SM_STATUS_NOT_SUPPORTED) return rc;
  }
  prune_tile_kernel<T, BF><<<stream_grid(ceil_div(m, 4) * ceil_div(k, 4), 256), 256, 0, st>>>((const T*)A_in, (T*)A_out, m, k, ld, vec_ok);
  return check_launch("prune_tile_kernel");
}

template <typename T>
static int launch_check(const void* A, size_t m, size_t k, size_t ld, int* d_valid, hipStream_t st) {
  if (!A || !d_valid || ld < k) {
    set_error("sm_prune24_check: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (hipMemsetAsync(d_valid, 0, sizeof(int), st) != hipSuccess) return check_launch("hipMemsetAsync");
  if (m == 0 || k == 0) return SM_STATUS_SUCCESS;
  const bool vec_ok = vec_ok_2d<T>(A, A, ld, 0);
  prune_check_kernel<T><<<stream_grid(m * ceil_div(k, 8), 256), 256, 0, st>>>((const T*)A, m, k, ld, vec_ok, d_valid);
  return check_launch("prune_check_kernel");
}

// the check kernel without the reset of the flag (prune_fused.hip: one flag over several batch matrices)
int prune_check_accumulate_u16(const void* A, size_t m, size_t k, size_t ld, int* d_valid, hipStream_t st) {
  if (m == 0 || k == 0) return SM_STATUS_SUCCESS;
  const bool vec_ok = vec_ok_2d<uint16_t>(A, A, ld, 0);
  prune_check_kernel<uint16_t><<<stream_grid(m * ceil_div(k, 8), 256), 256, 0, st>>>((const uint16_t*)A, m, k, ld, vec_ok, d_valid);
  return check_launch("prune_check_kernel");
}

int prune_check_accumulate_u32(const void* A, size_t m, size_t k, size_t ld, int* d_valid, hipStream_t st) {
  if (m == 0 || k == 0) return SM_STATUS_SUCCESS;
  const bool vec_ok = vec_ok_2d<uint32_t>(A, A, ld, 0);
  prune_check_kernel<uint32_t><<<stream_grid(m * ceil_div(k, 8), 256), 256, 0, st>>>((const uint32_t*)A, m, k, ld, vec_ok, d_valid);
  return check_launch("prune_check_kernel");
}

template <typename T>
static int launch_compress(const void* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob,
                           hipStream_t st) {
  if (!A || !blob || ld < k || !aligned16(blob)) {
    set_error("sm_compress24: invalid argument (blob must be 16-byte aligned)");
    return SM_STATUS_INVALID_VALUE;
  }
  const BlobLayout L = blob_layout(m, k, sizeof(T), batch);
  if (L.M == 0 || k == 0) return SM_STATUS_SUCCESS;
  // zero the alignment gaps so a blob is a pure function of its input
  const size_t vbytes = L.M * (L.kc / 2) * sizeof(T), mbytes = L.M * (L.kc / 8);
  if (L.meta_off > vbytes && hipMemsetAsync((char*)blob + vbytes, 0, L.meta_off - vbytes, st) != hipSuccess)
    return check_launch("hipMemsetAsync");
  if (L.total > L.meta_off + mbytes &&
      hipMemsetAsync((char*)blob + L.meta_off + mbytes, 0, L.total - L.meta_off - mbytes, st) != hipSuccess)
    return check_launch("hipMemsetAsync");
  const bool vec_ok = vec_ok_2d<T>(A, A, ld, strideA);
  const size_t items = L.M * (L.kc / 8);
  const size_t nunits = ceil_div(L.M, (size_t)8) * (L.kc / 64);
  if (vec_ok && L.kc == k && ld == k && (batch == 1 || strideA == m * ld) && nunits < 0xfffffff0ull) {
    static const bool nt = tuning_int("SM_COMPRESS_NT", 1) != 0;
    // two loads per lane (8 units per block) measured best on the ResNet-50 table (1.72 ms against 1.85 with 4,
    // 2.00 with 8, 1.84 with 1: profiles/sweep_r01_i_compress.txt)
    constexpr int NLD = 2;
    const unsigned nstage = (unsigned)(L.kc / 64), grid = (unsigned)ceil_div(nunits, (size_t)(4 * NLD));
    T* v = (T*)blob;
    unsigned char* mt = (unsigned char*)blob + L.meta_off;
    if (nt) compress_flat_kernel<T, true, NLD><<<grid, 256, 0, st>>>((const T*)A, L.M, k, nstage, (unsigned)nunits, v, mt);
    else compress_flat_kernel<T, false, NLD><<<grid, 256, 0, st>>>((const T*)A, L.M, k, nstage, (unsigned)nunits, v, mt);
    return check_launch("compress_flat_kernel");
  }
  if constexpr (sizeof(T) == 2) {
    // rows that are not whole 16-byte chunks (or k % 64 != 0), back to back: staged through LDS as one contiguous span
    const size_t span_bytes = 32 * ld * 2 + 32;
    if (aligned16(A) && (batch == 1 || strideA == m * ld) && span_bytes <= 48 * 1024 && L.kc / 64 <= 0xffffffu &&
        ceil_div(L.M, (size_t)32) < 0x7fffffffull) {
      compress_rowspan_f16_kernel<<<(unsigned)ceil_div(L.M, (size_t)32), 256, span_bytes, st>>>(
          (const uint16_t*)A, L.M, k, ld, (unsigned)(L.kc / 64), (uint16_t*)blob, (unsigned char*)blob + L.meta_off);
      return check_launch("compress_rowspan_f16_kernel");
    }
  }
  const unsigned grid = stream_grid(items, 256);
  compress_kernel<T><<<grid, 256, 0, st>>>((const T*)A, m, k, ld, strideA, L.kc, L.M, (T*)blob,
                                           (unsigned char*)blob + L.meta_off, vec_ok);
  return check_launch("compress_kernel");
}

template <typename T>
static int launch_decompress(const void* blob, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* A,
                             hipStream_t st) {
  if (!A || !blob || ld < k) {
    set_error("sm_decompress24: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  const BlobLayout L = blob_layout(m, k, sizeof(T), batch);
  if (L.M == 0 || k == 0) return SM_STATUS_SUCCESS;
  const bool vec_ok = vec_ok_2d<T>(A, A, ld, strideA);
  decompress_kernel<T><<<stream_grid(L.M * (L.kc / 8), 256), 256, 0, st>>>(
      (const T*)blob, (const unsigned char*)blob + L.meta_off, m, k, ld, strideA, L.kc, L.M, (T*)A, vec_ok);
  return check_launch("decompress_kernel");
}

}  // namespace sm

using namespace sm;

extern "C" {

int sm_sparsify_positional(void* weights, uint64_t* mask, size_t m, size_t n, size_t elt_bytes, size_t blk_m,
                           size_t blk_n, float sparsity_factor, sm_stream_t stream) {
  if (!weights || !mask || blk_m == 0 || blk_n == 0) {
    set_error("sm_sparsify_positional: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (elt_bytes) {
    case 2: return launch_sparsify<uint16_t>(weights, mask, m, n, blk_m, blk_n, sparsity_factor, st);
    case 4: return launch_sparsify<uint32_t>(weights, mask, m, n, blk_m, blk_n, sparsity_factor, st);
    case 8: return launch_sparsify<uint64_t>(weights, mask, m, n, blk_m, blk_n, sparsity_factor, st);
    default: set_error("sm_sparsify_positional: element size %zu not supported", elt_bytes); return SM_STATUS_INVALID_VALUE;
  }
}
int sm_sparsify_positional_f16(void* w, uint64_t* mask, size_t m, size_t n, float sf, sm_stream_t s) {
  return sm_sparsify_positional(w, mask, m, n, 2, 2, 2, sf, s);
}
int sm_sparsify_positional_f32(float* w, uint64_t* mask, size_t m, size_t n, float sf, sm_stream_t s) {
  return sm_sparsify_positional(w, mask, m, n, 4, 2, 2, sf, s);
}
int sm_sparsify_positional_f64(double* w, uint64_t* mask, size_t m, size_t n, float sf, sm_stream_t s) {
  return sm_sparsify_positional(w, mask, m, n, 8, 2, 2, sf, s);
}

int sm_prune24_f16(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, int alg, sm_stream_t s) {
  return launch_prune<uint16_t>(A_in, A_out, m, k, ld, alg, (hipStream_t)s);
}
/* bfloat16 (extension): the STRIP rule, the check, compress and decompress look at magnitude
 * BIT PATTERNS only, which order bfloat16 exactly as they order fp16 -- those entry points run the fp16 kernels as
 * they are; the TILE rule adds magnitudes and has its own instantiation. */
int sm_prune24_bf16(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, int alg, sm_stream_t s) {
  return launch_prune<uint16_t, true>(A_in, A_out, m, k, ld, alg, (hipStream_t)s);
}
int sm_prune24_check_bf16(const void* A, size_t m, size_t k, size_t ld, int* d_valid, sm_stream_t s) {
  return launch_check<uint16_t>(A, m, k, ld, d_valid, (hipStream_t)s);
}
int sm_compress24_bf16(const void* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob, sm_stream_t s) {
  return launch_compress<uint16_t>(A, m, k, ld, batch, strideA, blob, (hipStream_t)s);
}
int sm_decompress24_bf16(const void* blob, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* A, sm_stream_t s) {
  return launch_decompress<uint16_t>(blob, m, k, ld, batch, strideA, A, (hipStream_t)s);
}
int sm_prune24_f32(const float* A_in, float* A_out, size_t m, size_t k, size_t ld, int alg, sm_stream_t s) {
  return launch_prune<uint32_t>(A_in, A_out, m, k, ld, alg, (hipStream_t)s);
}
int sm_prune24_check_f16(const void* A, size_t m, size_t k, size_t ld, int* d_valid, sm_stream_t s) {
  return launch_check<uint16_t>(A, m, k, ld, d_valid, (hipStream_t)s);
}
int sm_prune24_check_f32(const float* A, size_t m, size_t k, size_t ld, int* d_valid, sm_stream_t s) {
  return launch_check<uint32_t>(A, m, k, ld, d_valid, (hipStream_t)s);
}

int sm_compress24_size(size_t m, size_t k, size_t elt_bytes, size_t batch, size_t* bytes) {
  if (!bytes || (elt_bytes != 1 && elt_bytes != 2 && elt_bytes != 4)) {
    set_error("sm_compress24_size: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  *bytes = blob_layout(m, k, elt_bytes, batch).total;
  return SM_STATUS_SUCCESS;
}
int sm_compress24_f16(const void* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob, sm_stream_t s) {
  return launch_compress<uint16_t>(A, m, k, ld, batch, strideA, blob, (hipStream_t)s);
}
int sm_compress24_f32(const float* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob, sm_stream_t s) {
  return launch_compress<uint32_t>(A, m, k, ld, batch, strideA, blob, (hipStream_t)s);
}
int sm_decompress24_f16(const void* blob, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* A, sm_stream_t s) {
  return launch_decompress<uint16_t>(blob, m, k, ld, batch, strideA, A, (hipStream_t)s);
}
int sm_decompress24_f32(const void* blob, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, float* A, sm_stream_t s) {
  return launch_decompress<uint32_t>(blob, m, k, ld, batch, strideA, A, (hipStream_t)s);
}

}  // extern "C"
