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

}  // extern "C"
