// spmm.hip -- unstructured SpMM entry points (row a6 of the hot-path table; secondary to the 2:4 path):
//   sm_spmm_bell_f32 : Blocked-ELL x dense, replaces the cusparseSpMM call of the reference's
//                      include/sparsify.me/spmm.hxx:57-67,107-110 (one launch instead of one host thread
//                      and stream per batch)
//   sm_spmm_coo_f32  : COO (one matrix shared by all batches) x strided dense batch, the intent of
//                      spmm.hxx:164-187
// Dense operands are column-major as the reference declares them.  Both kernels are HBM/L2-bound
// gathers: lanes run along the rows of C (contiguous in column-major), every lane walks its own row
// of A, and the B column it needs is small enough to stay in L1/L2.
#include "sm_common.h"

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
  for (size_t e = i; e + 1 < nnz; e += (size_t)gridDim.x * 256) bad |= rows[e] > rows[e + 1];
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(ws, 1);
}

constexpr int CSR_J = 16;
__global__ __launch_bounds__(256) void spmm_csr_kernel(size_t A_rows, size_t A_cols, size_t n, const int* __restrict__ ws,
                                                       const int* __restrict__ colsidx, const float* __restrict__ vals,
                                                       const float* __restrict__ B, float* __restrict__ C, float alpha,
                                                       float beta) {
  if (ws[0] != 0) return;  // unsorted input: the atomic kernel handles it
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

// atomic fallback gated on the flag (runs only when the rows were NOT sorted)
__global__ __launch_bounds__(256) void scale_if_unsorted_kernel(const int* ws, float* C, size_t count, float beta) {
  if (ws[0] == 0) return;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    C[i] = beta != 0.0f ? beta * C[i] : 0.0f;
}
__global__ __launch_bounds__(256) void spmm_coo_if_unsorted_kernel(const int* ws, size_t A_rows, size_t A_cols, size_t nnz,
                                                                   size_t n, const int* __restrict__ rows,
                                                                   const int* __restrict__ colsidx,
                                                                   const float* __restrict__ vals,
                                                                   const float* __restrict__ B, float* C, float alpha) {
  if (ws[0] == 0) return;
  const size_t e = blockIdx.x * (size_t)256 + threadIdx.x;
  if (e >= nnz) return;
  const size_t b = blockIdx.y / n, j = blockIdx.y % n;
  const size_t r = (size_t)rows[e], c = (size_t)colsidx[e];
  if (r >= A_rows || c >= A_cols) return;
  atomicAdd(C + b * A_rows * n + j * A_rows + r, alpha * vals[e] * B[b * A_cols * n + j * A_cols + c]);
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
  if (hipMemsetAsync(dense, 0, rows * cols * sizeof(float), st) != hipSuccess) return check_launch("hipMemsetAsync");
  bell_scatter_kernel<<<stream_grid(rows * ell_cols, 256), 256, 0, st>>>(values, column_indices, rows, cols, block_size, ell_cols, dense);
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
  if (hipMemsetAsync(dense, 0, batch * rows * cols * sizeof(float), st) != hipSuccess) return check_launch("hipMemsetAsync");
  dim3 grid(stream_grid(rows * ell_cols, 256), (unsigned)batch);
  bell_scatter_batched_kernel<<<grid, 256, 0, st>>>(d_vals, d_idx, rows, cols, block_size, ell_cols, dense);
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
  if (B_num_cols * num_batches > 65535 || num_batches > 65535 || A_nnz > 0x7fffffffull) {
    set_error("sm_spmm_coo_f32_ws: shape not supported");
    return SM_STATUS_NOT_SUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  int* ws = (int*)workspace;
  if (hipMemsetAsync(ws, 0, sizeof(int), st) != hipSuccess) return check_launch("hipMemsetAsync");
  coo_rowptr_kernel<<<(unsigned)ceil_div(A_num_rows + 1, 256), 256, 0, st>>>(rows, A_nnz, A_num_rows, ws);
  dim3 grid((unsigned)ceil_div(A_num_rows, 256), (unsigned)ceil_div(B_num_cols, CSR_J), (unsigned)num_batches);
  spmm_csr_kernel<<<grid, dim3(256), 0, st>>>(A_num_rows, A_num_cols, B_num_cols, ws, cols, vals, B, C, alpha, beta);
  scale_if_unsorted_kernel<<<stream_grid(count, 256), 256, 0, st>>>(ws, C, count, beta);
  if (A_nnz) {
    dim3 g2((unsigned)ceil_div(A_nnz, 256), (unsigned)(B_num_cols * num_batches));
    spmm_coo_if_unsorted_kernel<<<g2, dim3(256), 0, st>>>(ws, A_num_rows, A_num_cols, A_nnz, B_num_cols, rows, cols, vals, B, C, alpha);
  }
  return check_launch("spmm_csr_kernel");
}

}  // extern "C"
