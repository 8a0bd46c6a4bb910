// spmm.hip -- unstructured SpMM entry points (row a6 of the hot-path table; secondary to the 2:4 path):
//   sm_spmm_bell_f32 : Blocked-ELL x dense, replaces the cusparseSpMM call of the reference's
//                      include/sparsify.me/spmm.hxx:57-67,107-110 (one launch instead of one host thread
//                      and stream per batch)
//   sm_spmm_coo_f32  : COO (one matrix shared by all batches) x strided dense batch, the intent of
//                      spmm.hxx:164-187
// Dense operands are column-major as the reference declares them.  Both kernels are HBM/L2-bound
// gathers: lanes run along the rows of C (contiguous in column-major), every lane walks its own row
// of A, and the B column it needs is small enough to stay in L1/L2.
#include <cstdlib>

#include "sm_common.h"
#include "coo_fast.h"

namespace sm {

// C (rows x n, ldc = rows) = alpha * A_bell * B (cols x n, ldb = cols) + beta * C.
// One thread per (row, 8-column group).
constexpr int BELL_J = 8;
__global__ __launch_bounds__(256) void spmm_bell_kernel(const float* __restrict__ values,
                                                        const uint64_t* __restrict__ column_indices, size_t rows,
                                                        size_t cols, size_t block_size, size_t ell_cols,
                                                        const float* __restrict__ B, float* __restrict__ C, size_t n,
                                                        float alpha, float beta) {
  const size_t row = blockIdx.x * (size_t)256 + threadIdx.x;
  const size_t j0 = (size_t)blockIdx.y * BELL_J;
  if (row >= rows) return;
  const size_t bcols = ell_cols / block_size, nbc = cols / block_size;
  const size_t br = row / block_size;
  float acc[BELL_J];
#pragma unroll
  for (int j = 0; j < BELL_J; ++j) acc[j] = 0.0f;
  for (size_t e = 0; e < bcols; ++e) {
    const uint64_t bc = column_indices[br * bcols + e];
    if (bc >= nbc) continue;  // empty block
    for (size_t t = 0; t < block_size; ++t) {
      const float a = values[row * ell_cols + e * block_size + t];
      const size_t kk = bc * block_size + t;
#pragma unroll
      for (int j = 0; j < BELL_J; ++j)
        if (j0 + j < n) acc[j] = fmaf(a, B[(j0 + j) * cols + kk], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < BELL_J; ++j)
    if (j0 + j < n) {
      float* d = C + (j0 + j) * rows + row;
      *d = beta != 0.0f ? alpha * acc[j] + beta * *d : alpha * acc[j];
    }
}

// Blocked-ELL with a workspace: the blocks are scattered into a zeroed dense row-major A (rows x cols) and the
// product runs on the fp32 MFMA GEMM (gemm_f32.hip) -- with the reference's 50 %-dense 2 x 2 blocks that is
// 2x the necessary flops on the matrix cores instead of a gather-dot product on the VALU (100x faster here).
__global__ __launch_bounds__(256) void bell_scatter_kernel(const float* __restrict__ values,
                                                           const uint64_t* __restrict__ column_indices, size_t rows,
                                                           size_t cols, size_t block_size, size_t ell_cols,
                                                           float* __restrict__ dense) {
  const size_t total = rows * ell_cols;
  const size_t bcols = ell_cols / block_size, nbc = cols / block_size;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / ell_cols, ec = i - row * ell_cols, e = ec / block_size, t = ec - e * block_size;
    const uint64_t bc = column_indices[(row / block_size) * bcols + e];
    if (bc < nbc) dense[row * cols + bc * block_size + t] = values[i];
  }
}
// all batches in one grid: blockIdx.y = batch, value / index tables read through device pointer arrays
__global__ __launch_bounds__(256) void bell_scatter_batched_kernel(const float* const* __restrict__ values,
                                                                   const uint64_t* const* __restrict__ column_indices,
                                                                   size_t rows, size_t cols, size_t block_size,
                                                                   size_t ell_cols, float* __restrict__ dense) {
  const size_t total = rows * ell_cols;
  const size_t bcols = ell_cols / block_size, nbc = cols / block_size;
  const float* __restrict__ v = values[blockIdx.y];
  const uint64_t* __restrict__ ci = column_indices[blockIdx.y];
  float* __restrict__ d = dense + (size_t)blockIdx.y * rows * cols;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / ell_cols, ec = i - row * ell_cols, e = ec / block_size, t = ec - e * block_size;
    const uint64_t bc = ci[(row / block_size) * bcols + e];
    if (bc < nbc) d[row * cols + bc * block_size + t] = v[i];
  }
}
// Row-expand form of the scatter (cols * 4 B <= 48 KiB): one workgroup builds one dense row in LDS -- zero, scatter the
// row's ELL entries, write out with 16-byte stores -- so the dense workspace is written exactly once, coalesced, with
// no memset pass and no 64-bit division per element (the element-per-thread kernels above: 0.24 ms of the 0.54 ms
// Blocked-ELL product at 784 x 2304, b = 32).  blockIdx.y = batch (single-matrix form: pointer tables null).
__global__ __launch_bounds__(256) void bell_expand_rows_kernel(const float* const* __restrict__ values_tab,
                                                               const uint64_t* const* __restrict__ indices_tab,
                                                               const float* __restrict__ values1,
                                                               const uint64_t* __restrict__ indices1, unsigned rows, unsigned cols,
                                                               unsigned block_size, unsigned ell_cols, float* __restrict__ dense) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];
  const float* __restrict__ v = values_tab ? values_tab[blockIdx.y] : values1;
  const uint64_t* __restrict__ ci = indices_tab ? indices_tab[blockIdx.y] : indices1;
  const unsigned bcols = ell_cols / block_size, nbc = cols / block_size;
  float* __restrict__ d = dense + (size_t)blockIdx.y * rows * cols;
  for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
    for (unsigned c = threadIdx.x; c < cols; c += 256) rowbuf[c] = 0.0f;
    __syncthreads();
    const uint64_t* cirow = ci + (size_t)(row / block_size) * bcols;
    for (unsigned ec = threadIdx.x; ec < ell_cols; ec += 256) {
      const unsigned e = ec / block_size, t = ec - e * block_size;
      const uint64_t bc = cirow[e];
      if (bc < nbc) rowbuf[(unsigned)bc * block_size + t] = v[(size_t)row * ell_cols + ec];
    }
    __syncthreads();
    float* out = d + (size_t)row * cols;
    if ((cols & 3u) == 0 && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
      for (unsigned c = threadIdx.x; c < cols / 4; c += 256)
        __builtin_nontemporal_store(reinterpret_cast<const f4*>(rowbuf)[c], reinterpret_cast<f4*>(out) + c);
    } else {
      for (unsigned c = threadIdx.x; c < cols; c += 256) out[c] = rowbuf[c];
    }
    __syncthreads();
  }
}
// true when the row-expand kernel was launched (it needs the row in LDS)
static bool launch_bell_expand(const float* const* vt, const uint64_t* const* it, const float* v1, const uint64_t* i1, size_t rows,
                               size_t cols, size_t block_size, size_t ell_cols, size_t batch, float* dense, hipStream_t st) {
  if (cols * sizeof(float) > 48 * 1024 || rows > 0x7fffffffull || ell_cols > 0x7fffffffull || batch > 65535) return false;
  const unsigned gx = (unsigned)(rows < 65535 * 16 ? rows : 65535 * 16);
  bell_expand_rows_kernel<<<dim3(gx, (unsigned)batch), 256, cols * sizeof(float), st>>>(vt, it, v1, i1, (unsigned)rows, (unsigned)cols,
                                                                                         (unsigned)block_size, (unsigned)ell_cols, dense);
  return true;
}

int gemm_f32_colmajor_c_from_rowmajor_a(const float* Adense, const float* Bcm, float* Ccm, float* const* Cptrs,
                                        size_t m, size_t n, size_t k, size_t batch, float alpha, float beta,
                                        hipStream_t st);  // gemm_f32.hip

// C_b = beta * C_b (or 0) for every batch, then every (non-zero, column, batch) adds its product.
__global__ __launch_bounds__(256) void scale_kernel(float* C, size_t count, float beta) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    C[i] = beta != 0.0f ? beta * C[i] : 0.0f;
}
__global__ __launch_bounds__(256) void spmm_coo_kernel(size_t A_rows, size_t A_cols, size_t nnz, size_t n, size_t batches,
                                                       const int* __restrict__ rows, const int* __restrict__ colsidx,
                                                       const float* __restrict__ vals, const float* __restrict__ B,
                                                       float* C, float alpha) {
  // grid.x over non-zeros, grid.y over (batch, column); duplicates accumulate through the atomic
  const size_t e = blockIdx.x * (size_t)256 + threadIdx.x;
  if (e >= nnz) return;
  const size_t b = blockIdx.y / n, j = blockIdx.y % n;
  const size_t r = (size_t)rows[e], c = (size_t)colsidx[e];
  if (r >= A_rows || c >= A_cols) return;
  const float v = alpha * vals[e] * B[b * A_cols * n + j * A_cols + c];
  atomicAdd(C + b * A_rows * n + j * A_rows + r, v);
}

// ---- COO with a workspace: row-sorted input (the usual case: a row-major scan of a dense matrix) is turned
// into CSR row pointers by one binary search per row, and C is produced row-parallel with plain coalesced
// stores (lanes run along the rows of column-major C; every output is written exactly once: no atomics, the
// result is bitwise reproducible).  The same pass records whether the rows really are sorted; if not, the CSR
// kernel exits and the atomic kernel above does the work.  workspace: (A_rows + 2) ints.
__global__ __launch_bounds__(256) void coo_rowptr_kernel(const int* __restrict__ rows, size_t nnz, size_t A_rows,
                                                         int* __restrict__ ws) {
  int* row_ptr = ws + 1;  // ws[0] = "unsorted" flag (zeroed by the host-side memset node)
  const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  if (i <= A_rows) {  // row_ptr[i] = first e with rows[e] >= i
    size_t lo = 0, hi = nnz;
    while (lo < hi) {
      const size_t mid = (lo + hi) >> 1;
      if ((size_t)rows[mid] < i) lo = mid + 1; else hi = mid;
    }
    row_ptr[i] = (int)lo;
  }
  bool bad = false;
  // a row index outside [0, A_rows) breaks the monotonic sequence the search above relies on: such input goes to
  // the atomic kernel, which skips the bad entries one by one
  for (size_t e = i; e < nnz; e += (size_t)gridDim.x * 256) {
    const int r0 = rows[e];
    bad |= r0 < 0 || (size_t)r0 >= A_rows;
    if (e + 1 < nnz) bad |= r0 > rows[e + 1];
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(ws, 1);
}

// grid: the row-pointer search needs A_rows + 1 threads, the sortedness scan strides over the non-zeros with the whole
// grid -- sized by the rows alone it left 90 000 entries to one workgroup on a 196-row matrix (240 us)
static void launch_coo_rowptr(const int* rows, size_t nnz, size_t A_rows, int* ws, hipStream_t st) {
  size_t blocks = ceil_div(A_rows + 1 > nnz ? A_rows + 1 : nnz, (size_t)256);
  const size_t need = ceil_div(A_rows + 1, (size_t)256);
  if (blocks > 4096) blocks = 4096 > need ? 4096 : need;
  coo_rowptr_kernel<<<(unsigned)blocks, 256, 0, st>>>(rows, nnz, A_rows, ws);
}

constexpr int CSR_J = 16;
__global__ __launch_bounds__(256) void spmm_csr_kernel(size_t A_rows, size_t A_cols, size_t n, const int* __restrict__ ws,
                                                       const int* __restrict__ colsidx, const float* __restrict__ vals,
                                                       const float* __restrict__ B, float* __restrict__ C, float alpha,
                                                       float beta, int gate) {
  if (ws[0] != gate) return;  // unsorted input: the atomic kernel handles it
  const int* row_ptr = ws + 1;
  const size_t r = blockIdx.x * (size_t)256 + threadIdx.x;
  const size_t j0 = (size_t)blockIdx.y * CSR_J, b = blockIdx.z;
  if (r >= A_rows) return;
  const float* Bb = B + b * A_cols * n;
  float* Cb = C + b * A_rows * n;
  float acc[CSR_J];
#pragma unroll
  for (int j = 0; j < CSR_J; ++j) acc[j] = 0.0f;
  const int e0 = row_ptr[r], e1 = row_ptr[r + 1];
  for (int e = e0; e < e1; ++e) {
    const float a = vals[e];
    const size_t c = (size_t)colsidx[e];
    if (c >= A_cols) continue;
#pragma unroll
    for (int j = 0; j < CSR_J; ++j)
      if (j0 + j < n) acc[j] = fmaf(a, Bb[(j0 + j) * A_cols + c], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < CSR_J; ++j)
    if (j0 + j < n) {
      float* d = Cb + (j0 + j) * A_rows + r;
      *d = beta != 0.0f ? alpha * acc[j] + beta * *d : alpha * acc[j];
    }
}

// CSR with the dense operand in LDS.  Column j of column-major B_b (A_cols floats, contiguous) and column j of C_b are
// one sparse matrix-vector product; the batches only add columns (A is shared and the batch strides of B and C are
// exactly A_cols * n and A_rows * n), so the whole call is ONE product with NV = n * batches vectors.  A workgroup
// stages J = 8, 16 or 32 of those vectors in LDS, interleaved [A_cols][J], and its sixteen waves walk the rows: lanes
// (e = lane / (J/4), q = lane % (J/4)) take non-zero e of the 256/J the wave reads per step -- column index and value
// coalesced -- and the four vectors 4q .. 4q+3 with one ds_read_b128; partial sums are folded over e with xor-shuffles.  B is read from
// HBM exactly once, A comes from L2 (it is re-read by every workgroup), the gather happens in LDS: 0.6 TF/s -> see
// DESIGN.md for the measured rate against the thread-per-row kernel above, whose B gather went to global memory at a
// stride of A_cols floats.
constexpr int LDS_WAVES = 16;
template <int J>  // vectors per workgroup: 8, 16 or 32 (LDS = A_cols * J * 4 bytes)
__global__ __launch_bounds__(64 * LDS_WAVES) void spmm_csr_lds_kernel(size_t A_rows, size_t A_cols, size_t NV,
                                                                      const int* __restrict__ ws,
                                                                      const int* __restrict__ colsidx,
                                                                      const float* __restrict__ vals,
                                                                      const float* __restrict__ B, float* __restrict__ C,
                                                                      float alpha, float beta, int gate) {
  constexpr int QL = J / 4;    // lanes across the vectors (one 16-byte LDS read = four vectors each)
  constexpr int EL = 64 / QL;  // non-zeros a wave takes per step
  if (ws[0] != gate) return;  // 0: row-sorted input (1: the atomic kernel's case; 2 in the v2 path: columns unsorted within rows)
  extern __shared__ __attribute__((aligned(16))) float Xs[];  // [A_cols][J]
  const int* row_ptr = ws + 1;
  const size_t v0 = (size_t)blockIdx.x * J;
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // stage: a thread gathers element c of four vectors (each global read runs along c: coalesced) into one 16-byte store
  // (four iterations -- sixteen loads -- in flight per thread: the staging is a pure latency chain otherwise)
  const size_t n_it = A_cols * QL;
  for (size_t i0 = tid; i0 < n_it; i0 += 4 * 64 * LDS_WAVES) {
    f4 x[4];
#pragma unroll
    for (unsigned u = 0; u < 4; ++u) {
      const size_t i = i0 + u * 64 * LDS_WAVES;
      const size_t c = i / QL;
      const unsigned h = (unsigned)(i % QL);
#pragma unroll
      for (unsigned t = 0; t < 4; ++t) {
        const size_t v = v0 + 4u * h + t;
        x[u][t] = (i < n_it && v < NV) ? B[v * A_cols + c] : 0.0f;
      }
    }
#pragma unroll
    for (unsigned u = 0; u < 4; ++u) {
      const size_t i = i0 + u * 64 * LDS_WAVES;
      if (i < n_it) *reinterpret_cast<f4*>(Xs + (i / QL) * J + 4u * (unsigned)(i % QL)) = x[u];
    }
  }
  __syncthreads();
  const unsigned e = lane / QL, q = lane % QL;
  // blockIdx.y splits the rows (each split stages the same vectors again: B comes from L2 then)
  const size_t rows_per = (A_rows + gridDim.y - 1) / gridDim.y, r_begin = blockIdx.y * rows_per;
  const size_t r_end = r_begin + rows_per < A_rows ? r_begin + rows_per : A_rows;
  // A wave walks its rows G at a time: the column-index / value loads of one round of all G rows (8 G global loads) are
  // issued before the first of them is used.  With one row at a time a round's latency (L2, ~1-2 us under load) was
  // exposed once per 64 non-zeros and the kernel ran at 3 % of the HBM roofline: it is bound by that latency, not by
  // the LDS gather or the FMAs.
  constexpr int G = 4;
  for (size_t r0 = r_begin + (size_t)wave * G; r0 < r_end; r0 += (size_t)LDS_WAVES * G) {
    int e0[G], e1[G];
    int longest = 0;
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const size_t r = r0 + gi < r_end ? r0 + gi : r_end - 1;  // wave-uniform: scalar loads
      e0[gi] = row_ptr[r];
      e1[gi] = r0 + gi < r_end ? row_ptr[r + 1] : e0[gi];
      longest = e1[gi] - e0[gi] > longest ? e1[gi] - e0[gi] : longest;
    }
    f4 acc[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) acc[gi] = f4{0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < longest; base += 4 * EL) {
      int ci[G][4];
      float av[G][4];
#pragma unroll
      for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          // (round 6: these 32 predicated loads compile to one exec-mask branch each -- 62 s_and_saveexec + s_cbranch per round in the ISA.
          // Unconditional loads from a clamped index with the value zeroed by a select made the inner loop branch-free (32 loads, 32
          // ds_read_b128, 32 v_pk_fma_f32) and the kernel 3-5 % SLOWER on all 17 config-5 shapes (gpurun_out r06c vs r06a: 0.321 -> 0.339 ms
          // on 12544 x 64 x 576): the branches skip the loads of the ragged row tails, which is worth more than the scalar work costs.  Kept.)
          const int i = e0[gi] + base + EL * u + (int)e;
          const bool ok = i < e1[gi];
          ci[gi][u] = ok ? colsidx[i] : 0;
          av[gi][u] = ok ? vals[i] : 0.0f;
        }
#pragma unroll
      for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t c = (size_t)ci[gi][u] < A_cols ? (size_t)ci[gi][u] : 0;
          const float a = (size_t)ci[gi][u] < A_cols ? av[gi][u] : 0.0f;
          const f4 x = *reinterpret_cast<const f4*>(Xs + c * J + 4u * q);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[gi][t] = fmaf(a, x[t], acc[gi][t]);
        }
    }
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
#pragma unroll
      for (int off = QL; off < 64; off <<= 1)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[gi][t] += __shfl_xor(acc[gi][t], off, 64);
      if (e == 0 && r0 + gi < r_end) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const size_t v = v0 + 4u * q + t;
          if (v < NV) {
            float* d = C + v * A_rows + r0 + gi;
            *d = beta != 0.0f ? alpha * acc[gi][t] + beta * *d : alpha * acc[gi][t];
          }
        }
      }
    }
  }
}

template <int J>
static int launch_csr_lds(size_t A_rows, size_t A_cols, size_t nv, const int* ws, const int* cols, const float* vals,
                           const float* B, float* C, float alpha, float beta, hipStream_t st, int gate) {
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmm_csr_lds_kernel<J>), (144 * 1024), "spmm_csr_lds_kernel")) return rc;
  // enough workgroups for the chip: split the rows when there are few vector groups (>= 256 rows per split)
  const size_t groups = ceil_div(nv, (size_t)J);
  size_t rsplit = 1;
  while (groups * rsplit < 1024 && A_rows / (rsplit * 2) >= 256) rsplit *= 2;
  spmm_csr_lds_kernel<J><<<dim3((unsigned)groups, (unsigned)rsplit), 64 * LDS_WAVES, A_cols * J * sizeof(float), st>>>(A_rows, A_cols, nv, ws, cols, vals, B, C, alpha, beta, gate);
  return SM_STATUS_SUCCESS;
}

// atomic fallback gated on the flag (runs only when the rows were NOT sorted)
__global__ __launch_bounds__(256) void scale_if_unsorted_kernel(const int* ws, float* C, size_t count, float beta) {
  if ((ws[0] & 1) == 0) return;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    C[i] = beta != 0.0f ? beta * C[i] : 0.0f;
}
__global__ __launch_bounds__(256) void spmm_coo_if_unsorted_kernel(const int* ws, size_t A_rows, size_t A_cols, size_t nnz,
                                                                   size_t nv, const int* __restrict__ rows,
                                                                   const int* __restrict__ colsidx,
                                                                   const float* __restrict__ vals,
                                                                   const float* __restrict__ B, float* C, float alpha) {
  if ((ws[0] & 1) == 0) return;  // sorted input was served by a CSR kernel: this (small, grid-stride) launch costs nothing
  for (size_t v = blockIdx.y; v < nv; v += gridDim.y)
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < nnz; e += (size_t)gridDim.x * 256) {
      const size_t r = (size_t)rows[e], c = (size_t)colsidx[e];
      if (r >= A_rows || c >= A_cols) continue;
      atomicAdd(C + v * A_rows + r, alpha * vals[e] * B[v * A_cols + c]);
    }
}

// (A second form -- lanes along the dense vectors, the non-zeros wave-uniform in SGPRs through scalar loads of a pre-packed
// {LDS offset, value} stream, a row panel's partial sums in registers across 128-column chunks -- was built, tested bit for
// bit against the oracle and measured in round 2: 3 VALU / LDS instructions per 128 multiply-adds in the inner loop, and
// still 0.42-0.93 ms against 0.39-0.57 ms for the kernel above (profiles/spmm_probe_r02l_second_form.txt): with ~13
// non-zeros per (row, chunk) segment every segment pays two dependent scalar-load latencies, and lgkmcnt is shared
// between scalar loads and LDS reads.  Removed again (git history).)
// the row-parallel CSR kernels (vectors staged in LDS when they fit) followed by the gated atomic fallback; the CSR
// kernels run iff ws[0] == gate, the atomic ones iff ws[0] & 1
static int launch_csr_and_fallback(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches,
                                   const int* ws, const int* rows, const int* cols, const float* vals, const float* B, float* C,
                                   float alpha, float beta, hipStream_t st, int gate) {
  const size_t count = A_num_rows * B_num_cols * num_batches;
  const size_t nv = B_num_cols * num_batches, col_bytes = A_num_cols * sizeof(float);
  static const int lds_env = tuning_int("SM_SPMM_LDS", -1);  // tuning aid: 0 = off, 8/16/32 = J
  // as many vectors per workgroup as keep two workgroups on a CU (72 KB each): more FMAs per loaded non-zero
  int J = col_bytes * 32 <= 72 * 1024 ? 32 : (col_bytes * 16 <= 144 * 1024 ? 16 : 8);
  if (lds_env == 8 || lds_env == 16 || lds_env == 32) J = lds_env;
  if (lds_env != 0 && col_bytes * J <= 144 * 1024 && ceil_div(nv, (size_t)J) <= 0x7fffffffull) {
    int rc;
    if (J == 32) rc = launch_csr_lds<32>(A_num_rows, A_num_cols, nv, ws, cols, vals, B, C, alpha, beta, st, gate);
    else if (J == 16) rc = launch_csr_lds<16>(A_num_rows, A_num_cols, nv, ws, cols, vals, B, C, alpha, beta, st, gate);
    else rc = launch_csr_lds<8>(A_num_rows, A_num_cols, nv, ws, cols, vals, B, C, alpha, beta, st, gate);
    if (rc != SM_STATUS_SUCCESS) return rc;
  } else {
    dim3 grid((unsigned)ceil_div(A_num_rows, 256), (unsigned)ceil_div(B_num_cols, CSR_J), (unsigned)num_batches);
    spmm_csr_kernel<<<grid, dim3(256), 0, st>>>(A_num_rows, A_num_cols, B_num_cols, ws, cols, vals, B, C, alpha, beta, gate);
  }
  scale_if_unsorted_kernel<<<(unsigned)(ceil_div(count, (size_t)256) < 1024 ? ceil_div(count, (size_t)256) : 1024), 256, 0, st>>>(ws, C, count, beta);
  if (A_nnz) {
    // a capped grid with grid-stride loops: when the input is sorted (the flag is clear) these blocks exit at once,
    // and millions of empty blocks would cost more than the product itself
    const size_t gx = ceil_div(A_nnz, (size_t)256), gy = B_num_cols * num_batches;
    dim3 g2((unsigned)(gx < 64 ? gx : 64), (unsigned)(gy < 64 ? gy : 64));
    spmm_coo_if_unsorted_kernel<<<g2, dim3(256), 0, st>>>(ws, A_num_rows, A_num_cols, A_nnz, B_num_cols * num_batches, rows, cols, vals, B, C, alpha);
  }
  return check_launch("spmm_csr_kernel");
}



// ---------------------------------------------------------------------------------------------
// COO, PACKED form (sm_spmm_coo_f32_packed): the CSR-with-vectors-in-LDS kernel above with its per-entry bookkeeping
// moved into a re-ordering pass that runs once per call.  That kernel spends ~17 instructions per wave step (256
// multiply-adds): two loads, their predicates, the column clamps, the address, four FMAs, and per row a 16-shuffle
// reduction and 16 scattered 4-byte stores.  Here the rows are first copied into one stream of packed entries
// {LDS byte offset of the column's slab row, value}, each row padded to a multiple of 32 entries with {zero row, 0.0f}
// (an out-of-range column becomes such a pad: it is skipped, as before).  The product then needs per step one 8-byte
// load (the stream is contiguous across the rows a wave owns, so the next 32 entries are requested before the
// current ones are used), one add, one ds_read_b128 and two packed FMAs; no predicates, no clamps.  Per row the
// partial sums are folded with row-rotate DPP adds inside 16 lanes and two xor-shuffles across them; a lane keeps the
// totals of "its" row (row index mod the entries-per-step) and C is written every 64 / QL rows as 64-byte runs along
// the rows of column-major C.  Deterministic (no atomics; fixed summation order).  Taken iff the rows are sorted --
// decided on the device: ws[0] 0 -> 4 -- else the atomic kernels run.
// ---------------------------------------------------------------------------------------------
constexpr int PK_PAD = 32;  // entries: a multiple of the entries-per-step of every J

struct PkPlan {  // int offsets into the workspace
  size_t o_prow, o_ent, total_ints;
};
static PkPlan pk_plan(size_t m, size_t nnz) {
  PkPlan p;
  size_t o = m + 2;  // [0] flag, [1 .. m+1] row_ptr
  p.o_prow = o; o += m + 1;
  o = round_up(o, 4);
  p.o_ent = o; o += 2 * (nnz + (PK_PAD - 1) * m + 6 * PK_PAD);  // + slack: the kernel requests units past the last
  p.total_ints = o;
  return p;
}

// one workgroup: exclusive scan of the padded row lengths (a thread sums a contiguous run of rows, one scan of the 1024
// run totals, then the thread walks its run again)
__global__ __launch_bounds__(1024) void pk_scan_kernel(int* __restrict__ ws, int* __restrict__ prow, size_t A_rows) {
  if (ws[0] != 0) return;
  const int* row_ptr = ws + 1;
  __shared__ int part[1024];
  const unsigned tid = threadIdx.x;
  const size_t run = (A_rows + 1023) / 1024, r0 = tid * run, r1 = r0 + run < A_rows ? r0 + run : A_rows;
  int tot = 0;
  for (size_t r = r0; r < r1; ++r) tot += (row_ptr[r + 1] - row_ptr[r] + PK_PAD - 1) / PK_PAD * PK_PAD;
  part[tid] = tot;
  __syncthreads();
  for (unsigned d = 1; d < 1024; d <<= 1) {  // inclusive scan
    const int add = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  int at = part[tid] - tot;
  for (size_t r = r0; r < r1; ++r) {
    prow[r] = at;
    at += (row_ptr[r + 1] - row_ptr[r] + PK_PAD - 1) / PK_PAD * PK_PAD;
  }
  if (tid == 1023) {
    prow[A_rows] = part[1023];
    ws[0] = 4;  // the packed kernel takes the call
  }
}

// one wave per row: entries in input order, then the pads
__global__ __launch_bounds__(256) void pk_pack_kernel(const int* __restrict__ ws, const int* __restrict__ colsidx,
                                                      const float* __restrict__ vals, size_t A_rows, size_t A_cols,
                                                      unsigned row_bytes, const int* __restrict__ prow, u2* __restrict__ ent) {
  if (ws[0] != 4) return;
  const int* row_ptr = ws + 1;
  const unsigned lane = threadIdx.x & 63u;
  const size_t r = blockIdx.x * (size_t)4 + (threadIdx.x >> 6);
  if (r >= A_rows) return;
  const int e0 = row_ptr[r], n = row_ptr[r + 1] - e0, p0 = prow[r], pn = prow[r + 1] - p0;
  const unsigned zero_off = (unsigned)A_cols * row_bytes;
  for (int i = (int)lane; i < pn; i += 64) {
    u2 o = u2{zero_off, 0u};
    if (i < n) {
      const size_t c = (size_t)colsidx[e0 + i];
      if (c < A_cols) o = u2{(unsigned)c * row_bytes, __builtin_bit_cast(unsigned, vals[e0 + i])};
    }
    ent[p0 + i] = o;
  }
}

template <int N>
__device__ __forceinline__ float row_ror_add(float x) {  // x + (x rotated right by N inside each row of 16 lanes): one instruction
  float y;
  // (s_nop 1: a DPP read of a VGPR needs two wait states after the VALU write; inside inline asm nobody else inserts them)
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:%2 row_mask:0xf bank_mask:0xf" : "=v"(y) : "v"(x), "n"(N));
  return y;
}

template <int J>  // vectors per workgroup: 8, 16 or 32 (LDS = (A_cols + 1) * J * 4 bytes)
__global__ __launch_bounds__(64 * LDS_WAVES) void spmm_csr_packed_kernel(size_t A_rows, size_t A_cols, size_t NV,
                                                                          const int* __restrict__ ws, const int* __restrict__ prow,
                                                                          const u2* __restrict__ ent, const float* __restrict__ B,
                                                                          float* __restrict__ C, float alpha, float beta) {
  constexpr int QL = J / 4;        // lanes across the vectors (one 16-byte LDS read = four vectors each)
  constexpr int EL = 64 / QL;      // entries a wave takes per step, and rows per store group
  constexpr int SPU = PK_PAD / EL; // steps per 32-entry unit
  if (ws[0] != 4) return;
  extern __shared__ __attribute__((aligned(16))) float Xs[];  // [A_cols + 1][J], the last row zeros
  const size_t v0 = (size_t)blockIdx.x * J;
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // stage (as spmm_csr_lds_kernel): a thread gathers element c of four vectors into one 16-byte store
  const size_t n_it = A_cols * QL;
  for (size_t i0 = tid; i0 < n_it; i0 += 4 * 64 * LDS_WAVES) {
    f4 x[4];
#pragma unroll
    for (unsigned u = 0; u < 4; ++u) {
      const size_t i = i0 + u * 64 * LDS_WAVES;
      const size_t c = i / QL;
      const unsigned h = (unsigned)(i % QL);
#pragma unroll
      for (unsigned t = 0; t < 4; ++t) {
        const size_t v = v0 + 4u * h + t;
        x[u][t] = (i < n_it && v < NV) ? B[v * A_cols + c] : 0.0f;
      }
    }
#pragma unroll
    for (unsigned u = 0; u < 4; ++u) {
      const size_t i = i0 + u * 64 * LDS_WAVES;
      if (i < n_it) *reinterpret_cast<f4*>(Xs + (i / QL) * J + 4u * (unsigned)(i % QL)) = x[u];
    }
  }
  if (tid < (unsigned)J) Xs[A_cols * J + tid] = 0.0f;
  __syncthreads();
  const unsigned e = lane / QL, q = lane % QL;
  // blockIdx.y splits the rows; inside a split every wave owns a contiguous run of rows, i.e. one contiguous piece of
  // the entry stream, which it walks in 32-entry units, PD units requested ahead of the one in use
  constexpr int PD = 4;
  const size_t rows_per = (A_rows + gridDim.y - 1) / gridDim.y, r_begin = blockIdx.y * rows_per;
  const size_t r_end = r_begin + rows_per < A_rows ? r_begin + rows_per : A_rows;
  const size_t rpw = (r_end - r_begin + LDS_WAVES - 1) / LDS_WAVES;
  const size_t rw0 = r_begin + (size_t)wave * rpw;
  if (rw0 >= r_end) return;
  const size_t rw1 = rw0 + rpw < r_end ? rw0 + rpw : r_end;
  const char* xs = reinterpret_cast<const char*>(Xs) + 16u * q;
  const int pos0 = __builtin_amdgcn_readfirstlane(prow[rw0]);
  const int pos1 = __builtin_amdgcn_readfirstlane(prow[rw1]);
  const u2* eb = ent + pos0 + (int)e;
  const int units = (pos1 - pos0) / PK_PAD;
  u2 ring[PD][SPU];
#pragma unroll
  for (int d = 0; d < PD; ++d)
#pragma unroll
    for (int s = 0; s < SPU; ++s) ring[d][s] = eb[d * PK_PAD + s * EL];  // past the wave's piece: other rows or the slack
  // row bookkeeping in 32-bit scalars (the wave's rows are rw0 + ri)
  const int nrows = (int)(rw1 - rw0), rw0m = (int)(rw0 % EL);
  int ri = 0;
  auto row_end = [&](int i) {
    const size_t rr = rw0 + (size_t)i + 1;
    return __builtin_amdgcn_readfirstlane(prow[rr <= A_rows ? rr : A_rows]) - pos0;
  };
  int pend = row_end(0), pend1 = row_end(1), pend2 = row_end(2);  // in entries from pos0; two rows ahead
  f4 acc = f4{0.f, 0.f, 0.f, 0.f}, keep = f4{0.f, 0.f, 0.f, 0.f};
  const int bp16 = (int)((lane ^ 16u) * 4u), bp32 = (int)((lane ^ 32u) * 4u);  // ds_bpermute addresses of the xor partners
  auto flush = [&](int last) {  // rows [group base, last] of this wave are in `keep`, lane e <-> row % EL
    const int slot = (rw0m + last) % EL, i = last - slot + (int)e;  // wave-relative index of lane e's row
    if (i >= 0 && i <= last) {
      const size_t rr = rw0 + (size_t)i;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const size_t v = v0 + 4u * q + t;
        if (v < NV) {
          float* d = C + v * A_rows + rr;
          *d = beta != 0.0f ? alpha * keep[t] + beta * *d : alpha * keep[t];
        }
      }
    }
  };
  auto finish_rows = [&](int done) {  // every row that ends at `done` entries (empty rows included)
    while (ri < nrows && pend <= done) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float x = acc[t];
        if (QL <= 2) x = row_ror_add<2>(x);
        if (QL <= 4) x = row_ror_add<4>(x);
        x = row_ror_add<8>(x);
        x += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp16, __builtin_bit_cast(int, x)));
        x += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp32, __builtin_bit_cast(int, x)));
        acc[t] = x;
      }
      const int slot = (rw0m + ri) % EL;
      if ((int)e == slot) keep = acc;
      if (slot == EL - 1 || ri + 1 == nrows) flush(ri);
      acc = f4{0.f, 0.f, 0.f, 0.f};
      ++ri;
      pend = pend1;
      pend1 = pend2;
      pend2 = row_end(ri + 2);
    }
  };
  finish_rows(0);  // leading empty rows
  for (int u0 = 0; u0 < units; u0 += PD) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int u = u0 + d;
      if (u < units) {
        u2 cur[SPU];
#pragma unroll
        for (int s = 0; s < SPU; ++s) cur[s] = ring[d][s];
#pragma unroll
        for (int s = 0; s < SPU; ++s) ring[d][s] = eb[(u + PD) * PK_PAD + s * EL];
#pragma unroll
        for (int s = 0; s < SPU; ++s) {
          const f4 x = *reinterpret_cast<const f4*>(xs + cur[s][0]);
          const unsigned ab = cur[s][1];
          const float a = __builtin_bit_cast(float, ab);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = fmaf(a, x[t], acc[t]);
        }
        finish_rows((u + 1) * PK_PAD);
      }
    }
  }
}

template <int J>
static int launch_csr_packed(size_t A_rows, size_t A_cols, size_t nv, const int* ws, const PkPlan& p, const float* B, float* C,
                             float alpha, float beta, hipStream_t st) {
  const size_t lds = (A_cols + 1) * J * sizeof(float);
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&spmm_csr_packed_kernel<J>), (160 * 1024), "spmm_csr_packed_kernel")) return rc;
  const size_t groups = ceil_div(nv, (size_t)J);
  size_t rsplit = 1;
  while (groups * rsplit < 1024 && A_rows / (rsplit * 2) >= 256) rsplit *= 2;
  spmm_csr_packed_kernel<J><<<dim3((unsigned)groups, (unsigned)rsplit), 64 * LDS_WAVES, lds, st>>>(
      A_rows, A_cols, nv, ws, ws + p.o_prow, reinterpret_cast<const u2*>(ws + p.o_ent), B, C, alpha, beta);
  return check_launch("spmm_csr_packed_kernel");
}

}  // namespace sm

using namespace sm;

extern "C" {

int sm_spmm_bell_f32(const float* values, const uint64_t* column_indices, size_t rows, size_t cols, size_t block_size,
                     size_t ell_cols, const float* B, float* C, size_t n, float alpha, float beta, sm_stream_t stream) {
  if (!values || !column_indices || !B || !C || block_size == 0 || ell_cols % block_size != 0) {
    set_error("sm_spmm_bell_f32: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (rows == 0 || n == 0) return SM_STATUS_SUCCESS;
  const size_t gy = ceil_div(n, BELL_J);
  if (gy > 65535) {
    set_error("sm_spmm_bell_f32: n too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  dim3 grid((unsigned)ceil_div(rows, 256), (unsigned)gy);
  spmm_bell_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>(values, column_indices, rows, cols, block_size, ell_cols, B,
                                                                C, n, alpha, beta);
  return check_launch("spmm_bell_kernel");
}

int sm_spmm_coo_f32(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches,
                    const int* rows, const int* cols, const float* vals, const float* B, float* C, float alpha, float beta,
                    sm_stream_t stream) {
  if (!B || !C || (A_nnz && (!rows || !cols || !vals))) {
    set_error("sm_spmm_coo_f32: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  const size_t count = A_num_rows * B_num_cols * num_batches;
  if (count == 0) return SM_STATUS_SUCCESS;
  if (B_num_cols * num_batches > 65535) {
    set_error("sm_spmm_coo_f32: B_num_cols * num_batches exceeds 65535");
    return SM_STATUS_NOT_SUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  scale_kernel<<<stream_grid(count, 256), 256, 0, st>>>(C, count, beta);
  if (A_nnz) {
    dim3 grid((unsigned)ceil_div(A_nnz, 256), (unsigned)(B_num_cols * num_batches));
    spmm_coo_kernel<<<grid, dim3(256), 0, st>>>(A_num_rows, A_num_cols, A_nnz, B_num_cols, num_batches, rows, cols, vals, B, C, alpha);
  }
  return check_launch("spmm_coo_kernel");
}

int sm_spmm_bell_workspace_size(size_t rows, size_t cols, size_t* bytes) {
  if (!bytes) {
    set_error("sm_spmm_bell_workspace_size: null output");
    return SM_STATUS_INVALID_VALUE;
  }
  *bytes = round_up(rows * cols * sizeof(float), 256);
  return SM_STATUS_SUCCESS;
}

int sm_spmm_bell_f32_ws(const float* values, const uint64_t* column_indices, size_t rows, size_t cols, size_t block_size,
                        size_t ell_cols, const float* B, float* C, size_t n, float alpha, float beta, void* workspace,
                        sm_stream_t stream) {
  if (!workspace) return sm_spmm_bell_f32(values, column_indices, rows, cols, block_size, ell_cols, B, C, n, alpha, beta, stream);
  if (!values || !column_indices || !B || !C || block_size == 0 || ell_cols % block_size != 0) {
    set_error("sm_spmm_bell_f32_ws: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (rows == 0 || n == 0) return SM_STATUS_SUCCESS;
  if (rows > 0x7fffffffull || cols > 0x7fffffffull || n > 0x7fffffffull) {
    set_error("sm_spmm_bell_f32_ws: dimension exceeds 2^31-1");
    return SM_STATUS_NOT_SUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  float* dense = (float*)workspace;
  if (!launch_bell_expand(nullptr, nullptr, values, column_indices, rows, cols, block_size, ell_cols, 1, dense, st)) {
    if (hipMemsetAsync(dense, 0, rows * cols * sizeof(float), st) != hipSuccess) return check_launch("hipMemsetAsync");
    bell_scatter_kernel<<<stream_grid(rows * ell_cols, 256), 256, 0, st>>>(values, column_indices, rows, cols, block_size, ell_cols, dense);
  }
  if (check_launch("bell scatter") != SM_STATUS_SUCCESS) return SM_STATUS_LAUNCH_FAILED;
  return gemm_f32_colmajor_c_from_rowmajor_a(dense, B, C, nullptr, rows, n, cols, 1, alpha, beta, st);
}

int sm_spmm_bell_batched_workspace_size(size_t rows, size_t cols, size_t batch, size_t* bytes) {
  if (!bytes) {
    set_error("sm_spmm_bell_batched_workspace_size: null output");
    return SM_STATUS_INVALID_VALUE;
  }
  *bytes = round_up(batch * rows * cols * sizeof(float), 256) + round_up(3 * batch * sizeof(void*), 256);
  return SM_STATUS_SUCCESS;
}

int sm_spmm_bell_batched_f32(const float* const* values, const uint64_t* const* column_indices, size_t rows, size_t cols,
                             size_t block_size, size_t ell_cols, const float* B, float* const* C, size_t n, size_t batch,
                             float alpha, float beta, void* workspace, sm_stream_t stream) {
  if (!values || !column_indices || !B || !C || !workspace || block_size == 0 || ell_cols % block_size != 0) {
    set_error("sm_spmm_bell_batched_f32: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  if (rows == 0 || n == 0 || batch == 0) return SM_STATUS_SUCCESS;
  if (rows > 0x7fffffffull || cols > 0x7fffffffull || n > 0x7fffffffull || batch > 65535) {
    set_error("sm_spmm_bell_batched_f32: dimension out of range");
    return SM_STATUS_NOT_SUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  float* dense = (float*)workspace;
  const size_t dense_bytes = round_up(batch * rows * cols * sizeof(float), 256);
  char* tables = (char*)workspace + dense_bytes;
  const float** d_vals = (const float**)tables;
  const uint64_t** d_idx = (const uint64_t**)(tables + batch * sizeof(void*));
  float** d_c = (float**)(tables + 2 * batch * sizeof(void*));
  // the tables are host arrays of device pointers: three small synchronous-staged copies onto the stream
  if (hipMemcpyAsync(d_vals, values, batch * sizeof(void*), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_idx, column_indices, batch * sizeof(void*), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_c, C, batch * sizeof(void*), hipMemcpyHostToDevice, st) != hipSuccess)
    return check_launch("hipMemcpyAsync(pointer tables)");
  if (!launch_bell_expand(d_vals, d_idx, nullptr, nullptr, rows, cols, block_size, ell_cols, batch, dense, st)) {
    if (hipMemsetAsync(dense, 0, batch * rows * cols * sizeof(float), st) != hipSuccess) return check_launch("hipMemsetAsync");
    dim3 grid(stream_grid(rows * ell_cols, 256), (unsigned)batch);
    bell_scatter_batched_kernel<<<grid, 256, 0, st>>>(d_vals, d_idx, rows, cols, block_size, ell_cols, dense);
  }
  if (check_launch("bell scatter") != SM_STATUS_SUCCESS) return SM_STATUS_LAUNCH_FAILED;
  return gemm_f32_colmajor_c_from_rowmajor_a(dense, B, nullptr, d_c, rows, n, cols, batch, alpha, beta, st);
}

int sm_spmm_coo_workspace_size(size_t A_num_rows, size_t* bytes) {
  if (!bytes) {
    set_error("sm_spmm_coo_workspace_size: null output");
    return SM_STATUS_INVALID_VALUE;
  }
  *bytes = (A_num_rows + 2) * sizeof(int);
  return SM_STATUS_SUCCESS;
}

int sm_spmm_coo_f32_ws(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches,
                       const int* rows, const int* cols, const float* vals, const float* B, float* C, float alpha,
                       float beta, void* workspace, sm_stream_t stream) {
  if (!workspace) return sm_spmm_coo_f32(A_num_rows, A_num_cols, A_nnz, B_num_cols, num_batches, rows, cols, vals, B, C, alpha, beta, stream);
  if (!B || !C || (A_nnz && (!rows || !cols || !vals))) {
    set_error("sm_spmm_coo_f32_ws: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  const size_t count = A_num_rows * B_num_cols * num_batches;
  if (count == 0) return SM_STATUS_SUCCESS;
  // (the CSR kernels index their vector groups by grid x and the fallback by (rows, column groups, batches): no limit on
  //  B_num_cols * num_batches -- 196 x 2048 x 512 at b = 32 has 65 536 vectors; round 4)
  if (ceil_div(B_num_cols, (size_t)CSR_J) > 65535 || num_batches > 65535 || A_nnz > 0x7fffffffull) {
    set_error("sm_spmm_coo_f32_ws: shape not supported");
    return SM_STATUS_NOT_SUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  int* ws = (int*)workspace;
  if (hipMemsetAsync(ws, 0, sizeof(int), st) != hipSuccess) return check_launch("hipMemsetAsync");
  launch_coo_rowptr(rows, A_nnz, A_num_rows, ws, st);
  return launch_csr_and_fallback(A_num_rows, A_num_cols, A_nnz, B_num_cols, num_batches, ws, rows, cols, vals, B, C, alpha, beta, st, 0);
}


int sm_spmm_coo_packed_workspace_size(size_t A_num_rows, size_t A_nnz, size_t* bytes) {
  if (!bytes) {
    set_error("sm_spmm_coo_packed_workspace_size: null output");
    return SM_STATUS_INVALID_VALUE;
  }
  *bytes = round_up(pk_plan(A_num_rows, A_nnz).total_ints * sizeof(int), 256);
  return SM_STATUS_SUCCESS;
}

}  // extern "C" (re-opened after the dense-MFMA form's device code)

// ---------------------------------------------------------------------------------------------
// COO, DENSE-MFMA form (sm_spmm_coo_f32_fast; round 3; an explicit opt-in, NOT what strided_coo calls).  Every LDS-gather
// formulation of this product is bounded near 50 us on the 784 x 256 x 2304 layer by its LDS data reads alone and pays
// bank conflicts on top.  At 10 % density the matrix pipe does it faster DENSE: column j of column-major
// B_b is one contiguous k-vector, so with all batches the product is the row-major GEMM
//     C^T [(n b) x m] = Bcat^T [(n b) x k] * A^T [k x m]          (C_b column-major IS C^T's rows, ld = m)
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation.  Operands: Bcat^T rounded once to fp16 (relative error <= 2^-11 per
// element: inside north_star's 1e-3 of sum|a b|, asserted in the test); A^T scattered dense in fp32 (duplicates add) and
// split exactly into two fp16 planes hi + lo (|error| <= 2^-22 |a|), stacked along k so that ONE launch accumulates
// Bcat16 * hi + Bcat16 * lo (gemm_f16.hip: gemm_f16_f32out, the A operand's k wraps).  2 x the dense flops at fp16 MFMA rate
// instead of 1 x at fp32 rate (16 x slower).  Inputs beyond fp16's range (|x| > 65504) overflow: stated in the header.
// Workspace: (n b k) fp16 + (k m) fp32 + 2 (k m) fp16.  Needs k % 64 == 0, m % 4 == 0, m >= 8; else NOT_SUPPORTED.
// ---------------------------------------------------------------------------------------------
namespace sm {
int gemm_f16_f32out(const void* A, const void* B2, float* C, size_t M, size_t N, size_t K, size_t lda, size_t ldb, size_t ldc, float alpha,
                    float beta, hipStream_t st, const float* alpha_dev, const int* skip_flag);  // gemm_f16.hip

// max |a| over all values of A (exact) and max |b| over a SAMPLE of the dense operand: `nchunks` runs of 1024 contiguous floats
// spread evenly over it (whole cache lines: a strided element sample costs a line per element -- 35 us for a 10^6-element sample,
// measured in round 4's first form -- where these 4 MB cost 2-3)
// (round 5: 128 workgroups of 1024 threads, every thread's loads independent of one another -- four in flight -- and ONE pair of atomics per
// workgroup: the first form's 1024 wave-level atomics on two addresses and its dependent loads cost 26-28 us per call, profiles/coo_kernels_r05ab.txt)
__global__ __launch_bounds__(1024) void coo_fast_scan_kernel(const float* __restrict__ vals, size_t nnz, const float* __restrict__ B, size_t nb, size_t nchunks,
                                                             size_t chunk_step, CooFastHdr* hdr) {
  __shared__ unsigned red[2][16];
  unsigned ma = 0u, mb = 0u;
  const size_t t0 = blockIdx.x * (size_t)1024 + threadIdx.x, nt = (size_t)gridDim.x * 1024;
  for (size_t i = t0; i < nnz; i += 4 * nt) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i + j * nt < nnz ? vals[i + j * nt] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned u = __builtin_bit_cast(unsigned, v[j]) & 0x7fffffffu;
      ma = u > ma ? u : ma;
    }
  }
  const size_t units = nchunks * 256;  // (chunk, 16-byte piece of it)
  for (size_t u0 = t0; u0 < units; u0 += 4 * nt) {
    f4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t u = u0 + j * nt, i = (u >> 8) * chunk_step + 4u * (u & 255u);  // chunk_step % 4 == 0, B 16-byte aligned
      v[j] = u < units && i + 4 <= nb ? *reinterpret_cast<const f4*>(B + i) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ve = v[j][e];  // (a named copy: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 with this compiler)
        const unsigned u = __builtin_bit_cast(unsigned, ve) & 0x7fffffffu;
        mb = u > mb ? u : mb;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned xa = (unsigned)__shfl_xor((int)ma, o), xb = (unsigned)__shfl_xor((int)mb, o);
    ma = xa > ma ? xa : ma;
    mb = xb > mb ? xb : mb;
  }
  if ((threadIdx.x & 63u) == 0) {
    red[0][threadIdx.x >> 6] = ma;
    red[1][threadIdx.x >> 6] = mb;
  }
  __syncthreads();
  if (threadIdx.x < 32u) {
    unsigned x = red[threadIdx.x >> 4][threadIdx.x & 15u];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const unsigned y = (unsigned)__shfl_xor((int)x, o);
      x = y > x ? y : x;
    }
    if ((threadIdx.x & 15u) == 0 && x) atomicMax(threadIdx.x == 0 ? &hdr->max_a : &hdr->max_b, x);
  }
}
__global__ __launch_bounds__(256) void f32_to_f16_scaled_kernel(const float* __restrict__ in, _Float16* __restrict__ out, size_t n8, CooFastHdr* hdr) {
  const float sc = coo_fast_pow2(coo_fast_scale_exp(hdr->max_b, 12));
  bool bad = false;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4*>(in) + 2 * i);
    const f4 c = __builtin_nontemporal_load(reinterpret_cast<const f4*>(in) + 2 * i + 1);
    typedef _Float16 hv8 __attribute__((ext_vector_type(8)));
    const float x[8] = {a[0] * sc, a[1] * sc, a[2] * sc, a[3] * sc, c[0] * sc, c[1] * sc, c[2] * sc, c[3] * sc};
#pragma unroll
    for (int j = 0; j < 8; ++j) bad |= coo_fast_out_of_range(x[j]);
    const hv8 o = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3], (_Float16)x[4], (_Float16)x[5], (_Float16)x[6], (_Float16)x[7]};
    *(reinterpret_cast<hv8*>(out) + i) = o;
  }
  if (__any(bad) && (threadIdx.x & 63u) == 0) atomicOr(&hdr->flag, 1);
}
__global__ __launch_bounds__(256) void coo_scatter_dense_kernel(const int* __restrict__ rows, const int* __restrict__ cols, const float* __restrict__ vals,
                                                                size_t nnz, size_t A_rows, size_t A_cols, float* __restrict__ AT /*[cols][rows]*/) {
  for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < nnz; e += (size_t)gridDim.x * 256) {
    const size_t r = (size_t)rows[e], c = (size_t)cols[e];
    if (r < A_rows && c < A_cols) atomicAdd(AT + c * A_rows + r, vals[e]);  // an out-of-range coordinate is skipped, as in the other COO forms
  }
}
__global__ __launch_bounds__(256) void split_f16x2_scaled_kernel(const float* __restrict__ in, _Float16* __restrict__ hi, _Float16* __restrict__ lo, size_t n,
                                                                 CooFastHdr* hdr) {
  const int xa = coo_fast_scale_exp(hdr->max_a, 13);
  const float sc = coo_fast_pow2(xa);
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the matrix kernel, launched after this one, reads them
    hdr->inv_scale[0] = coo_fast_pow2(-xa);
    hdr->inv_scale[1] = coo_fast_pow2(-coo_fast_scale_exp(hdr->max_b, 12));
  }
  bool bad = false;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float x = in[i] * sc;
    bad |= coo_fast_out_of_range(x);   // the hi plane carries the 2^-11 bound; lo only refines it (A exact to 2^-22 where lo is normal too)
    const _Float16 h = (_Float16)x;
    hi[i] = h;
    lo[i] = (_Float16)(x - (float)h);
  }
  if (__any(bad) && (threadIdx.x & 63u) == 0) atomicOr(&hdr->flag, 1);
}
}  // namespace sm

namespace sm {
void coo_fast_scan(const float* vals, size_t nnz, const float* B, size_t nB, CooFastHdr* hdr, hipStream_t st) {
  // <= 1024 chunks of 1024 floats, evenly spread (a dense operand of <= 1 M elements is scanned whole)
  const size_t nchunks = nB / 1024 < 1024 ? (nB + 1023) / 1024 : 1024;
  const size_t chunk_step = nchunks >= 1024 ? (nB / nchunks) & ~(size_t)3 : 1024;
  coo_fast_scan_kernel<<<128, 1024, 0, st>>>(vals, nnz, B, nB, nchunks, chunk_step, hdr);
}
}  // namespace sm

static bool coo_fast_sizes(size_t m, size_t k, size_t n, size_t b, size_t* b16, size_t* at32, size_t* aop) {
  size_t nv, e;
  if (__builtin_mul_overflow(n, b, &nv) || __builtin_mul_overflow(nv, k, &e) || __builtin_mul_overflow(e, (size_t)2, b16) ||
      __builtin_mul_overflow(k, m, &e) || __builtin_mul_overflow(e, (size_t)4, at32) || __builtin_mul_overflow(e, (size_t)4, aop))
    return false;
  *b16 = sm::round_up(*b16, 256); *at32 = sm::round_up(*at32, 256); *aop = sm::round_up(*aop, 256);
  return *b16 < ((size_t)1 << 62) && *at32 < ((size_t)1 << 62);
}

extern "C" int sm_spmm_coo_fast_workspace_size(size_t A_num_rows, size_t A_num_cols, size_t B_num_cols, size_t num_batches, size_t* bytes) {
  if (!bytes) return SM_STATUS_INVALID_VALUE;
  size_t b16, at32, aop;
  if (!coo_fast_sizes(A_num_rows, A_num_cols, B_num_cols, num_batches, &b16, &at32, &aop)) {
    sm::set_error("sm_spmm_coo_fast_workspace_size: the operand sizes overflow size_t");
    *bytes = 0;
    return SM_STATUS_NOT_SUPPORTED;
  }
  *bytes = sm::COO_FAST_HDR_BYTES + b16 + at32 + aop;
  // the sparse-matrix-instruction form (spmm_coo_smfmac.hip) lays the same workspace out its own way
  size_t nv = 0;
  if (!__builtin_mul_overflow(B_num_cols, num_batches, &nv)) {
    const size_t sp = sm::coo_smfmac_workspace(A_num_rows, A_num_cols, nv);
    if (sp > *bytes) *bytes = sp;
  }
  return SM_STATUS_SUCCESS;
}

extern "C" int sm_spmm_coo_fast_form(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches, float beta) {
  size_t nv = 0, need = 0;
  if (__builtin_mul_overflow(B_num_cols, num_batches, &nv) || nv == 0 || nv > 0x7fffffffull || A_num_rows == 0 ||
      sm_spmm_coo_fast_workspace_size(A_num_rows, A_num_cols, B_num_cols, num_batches, &need) != SM_STATUS_SUCCESS)
    return 0;
  static const int smfmac_env = sm::tuning_int("SM_COO_SMFMAC", 1);  // (tuning builds only; the product never reads the environment)
  if (smfmac_env && sm::coo_smfmac_takes(A_num_rows, A_num_cols, A_nnz, nv, nullptr, nullptr, beta)) return 2;
  return A_num_cols != 0 && A_num_cols % 64 == 0 && A_num_rows % 4 == 0 && A_num_rows >= 8 ? 1 : 0;
}

extern "C" int sm_spmm_coo_f32_fast(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches,
                                    const int* rows, const int* cols, const float* vals, const float* B, float* C, float alpha,
                                    float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  using namespace sm;
  const size_t m = A_num_rows, k = A_num_cols, nv = B_num_cols * num_batches;
  if (nv == 0 || m == 0) return SM_STATUS_SUCCESS;
  if (!B || !C || (A_nnz && (!rows || !cols || !vals)) || !workspace || !aligned16(workspace)) {
    set_error("sm_spmm_coo_f32_fast: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  size_t need = 0;
  const int rs = sm_spmm_coo_fast_workspace_size(m, k, B_num_cols, num_batches, &need);
  // round 5: beta == 0 and a sparse enough A -> the product on the SPARSE matrix instruction (one prepared 2:4 image of A + the few entries
  // that do not fit it, the dense operand converted in the loader), any k
  static const int smfmac_env = tuning_int("SM_COO_SMFMAC", 1);  // tuning aid: 0 = always the dense-MFMA pipeline
  if (rs == SM_STATUS_SUCCESS && workspace_bytes >= need && nv <= 0x7fffffffull && smfmac_env && coo_smfmac_takes(m, k, A_nnz, nv, B, C, beta))
    return coo_smfmac_product(m, k, A_nnz, nv, rows, cols, vals, B, C, alpha, workspace, (hipStream_t)stream);
  if (rs != SM_STATUS_SUCCESS || k == 0 || k % 64 != 0 || m % 4 != 0 || m < 8 || !aligned16(B) || !aligned16(C) || workspace_bytes < need || nv > 0x7fffffffull) {
    set_error("sm_spmm_coo_f32_fast: needs cols %% 64 == 0 (cols > 0), rows %% 4 == 0, 16-byte aligned B and C and the workspace of sm_spmm_coo_fast_workspace_size "
              "(use sm_spmm_coo_f32_packed)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  CooFastHdr* hdr = (CooFastHdr*)ws;                       // [header | A^T fp32 | dense operand fp16 | A planes fp16]
  float* AT = (float*)(ws + COO_FAST_HDR_BYTES);
  _Float16* B16 = (_Float16*)((char*)AT + round_up(k * m * 4, 256));
  _Float16* Aop = (_Float16*)((char*)B16 + round_up(nv * k * 2, 256));
  if (hipMemsetAsync(ws, 0, COO_FAST_HDR_BYTES + k * m * 4, st) != hipSuccess) return check_launch("hipMemsetAsync");  // header and A^T in one node
  const size_t nB = nv * k;
  coo_fast_scan(vals, A_nnz, B, nB, hdr, st);
  const size_t n8 = nB / 8;  // k % 64 == 0
  f32_to_f16_scaled_kernel<<<stream_grid(n8, 256), 256, 0, st>>>(B, B16, n8, hdr);
  if (A_nnz) coo_scatter_dense_kernel<<<stream_grid(A_nnz, 256), 256, 0, st>>>(rows, cols, vals, A_nnz, m, k, AT);
  split_f16x2_scaled_kernel<<<stream_grid(k * m, 256), 256, 0, st>>>(AT, Aop, Aop + k * m, k * m, hdr);
  if (const int rc = check_launch("sm_spmm_coo_f32_fast: operand preparation")) return rc;
  return gemm_f16_f32out(B16, Aop, C, nv, m, k, k, m, m, alpha, beta, st, hdr->inv_scale, &hdr->flag);
}

extern "C" int sm_spmm_coo_fast_flag(const void* workspace, int* host_flag, sm_stream_t stream) {
  if (!workspace || !host_flag) return SM_STATUS_INVALID_VALUE;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(host_flag, workspace, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return sm::check_launch("sm_spmm_coo_fast_flag");
  return SM_STATUS_SUCCESS;
}

extern "C" {

int sm_spmm_coo_f32_packed(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches,
                           const int* rows, const int* cols, const float* vals, const float* B, float* C, float alpha,
                           float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  const PkPlan p = pk_plan(A_num_rows, A_nnz);
  const size_t nv = B_num_cols * num_batches, row_bytes_max = (A_num_cols + 1) * sizeof(float);
  // as many vectors per workgroup as the slab (columns + a zero row) leaves room for: 32 while two workgroups share a CU
  int J = row_bytes_max * 32 <= 80 * 1024 ? 32 : (row_bytes_max * 16 <= 160 * 1024 ? 16 : 8);
  static const int pk_j_env = tuning_int("SM_SPMM_PK_J", 0);  // tuning aid: 8 / 16 / 32 vectors per workgroup
  const bool forced = pk_j_env == 8 || pk_j_env == 16 || pk_j_env == 32;
  if (forced) J = pk_j_env;
  const bool can = workspace && workspace_bytes >= p.total_ints * sizeof(int) && A_nnz > 0 && aligned16(workspace) &&
                   (J >= 16 || forced) /* 8 vectors per workgroup: measured slower than the row-pointer form (445 vs 339 us) */ &&
                   row_bytes_max * J <= 160 * 1024 && A_nnz + PK_PAD * (A_num_rows + 2) <= 0x7fffffffull &&
                   A_num_rows <= 0x7ffffff0ull && ceil_div(nv, (size_t)J) <= 0x7fffffffull;
  if (!can) {
    size_t small = 0;
    (void)sm_spmm_coo_workspace_size(A_num_rows, &small);
    return sm_spmm_coo_f32_ws(A_num_rows, A_num_cols, A_nnz, B_num_cols, num_batches, rows, cols, vals, B, C, alpha, beta,
                              workspace && workspace_bytes >= small ? workspace : nullptr, stream);
  }
  if (!B || !C || !rows || !cols || !vals) {
    set_error("sm_spmm_coo_f32_packed: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  const size_t count = A_num_rows * nv;
  if (count == 0) return SM_STATUS_SUCCESS;
  hipStream_t st = (hipStream_t)stream;
  int* ws = (int*)workspace;
  if (hipMemsetAsync(ws, 0, sizeof(int), st) != hipSuccess) return check_launch("hipMemsetAsync");
  launch_coo_rowptr(rows, A_nnz, A_num_rows, ws, st);
  pk_scan_kernel<<<1, 1024, 0, st>>>(ws, ws + p.o_prow, A_num_rows);
  pk_pack_kernel<<<(unsigned)ceil_div(A_num_rows, (size_t)4), 256, 0, st>>>(ws, cols, vals, A_num_rows, A_num_cols, (unsigned)(J * sizeof(float)),
                                                                            ws + p.o_prow, reinterpret_cast<u2*>(ws + p.o_ent));
  if (check_launch("sm_spmm_coo_f32_packed: re-ordering") != SM_STATUS_SUCCESS) return SM_STATUS_LAUNCH_FAILED;
  int rc;
  if (J == 32) rc = launch_csr_packed<32>(A_num_rows, A_num_cols, nv, ws, p, B, C, alpha, beta, st);
  else if (J == 16) rc = launch_csr_packed<16>(A_num_rows, A_num_cols, nv, ws, p, B, C, alpha, beta, st);
  else rc = launch_csr_packed<8>(A_num_rows, A_num_cols, nv, ws, p, B, C, alpha, beta, st);
  if (rc != SM_STATUS_SUCCESS) return rc;
  // rows not sorted (flag odd): the atomic kernels
  scale_if_unsorted_kernel<<<(unsigned)(ceil_div(count, (size_t)256) < 1024 ? ceil_div(count, (size_t)256) : 1024), 256, 0, st>>>(ws, C, count, beta);
  const size_t gx = ceil_div(A_nnz, (size_t)256);
  dim3 g2((unsigned)(gx < 64 ? gx : 64), (unsigned)(nv < 64 ? nv : 64));
  spmm_coo_if_unsorted_kernel<<<g2, dim3(256), 0, st>>>(ws, A_num_rows, A_num_cols, A_nnz, nv, rows, cols, vals, B, C, alpha);
  return check_launch("sm_spmm_coo_f32_packed");
}

}  // extern "C"
