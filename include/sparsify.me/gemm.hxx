// gemm.hxx -- sparsifyme::batched::gemm: dense batched GEMM, the metric's denominator.
// Signature and semantics of the reference's include/sparsify.me/gemm.hxx:25-36 (specialised there
// on cublas{H,S,D}gemmBatched, :38/:91/:144): column-major, lda = m, ldb = k, ldc = m, device
// arrays of device pointers, returns the elapsed milliseconds of the GEMM alone and blocks until
// it has finished.  Here the call lands on the hand-written MFMA kernels of libsparsifyme.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstddef>
#include <iostream>

#include <sparsifyme.h>
#include <sparsify.me/util/trace.hxx>
#include <sparsify.me/util/util.hxx>

namespace sparsifyme {

// stands where the reference's signatures say cublasOperation_t / cusparseOperation_t
enum operation_t { N = SM_OP_N, T = SM_OP_T };

namespace batched {
namespace detail {
inline int gemm_call(_Float16** A, _Float16** B, _Float16** C, std::size_t m, std::size_t n, std::size_t k, std::size_t b,
                     int ta, int tb, float alpha, float beta) {
  return sm_gemm_batched_f16((const void* const*)A, (const void* const*)B, (void* const*)C, m, n, k, b, ta, tb, alpha, beta, nullptr);
}
inline int gemm_call(__half** A, __half** B, __half** C, std::size_t m, std::size_t n, std::size_t k, std::size_t b, int ta,
                     int tb, __half alpha, __half beta) {
  return sm_gemm_batched_f16((const void* const*)A, (const void* const*)B, (void* const*)C, m, n, k, b, ta, tb,
                             __half2float(alpha), __half2float(beta), nullptr);
}
inline int gemm_call(float** A, float** B, float** C, std::size_t m, std::size_t n, std::size_t k, std::size_t b, int ta,
                     int tb, float alpha, float beta) {
  return sm_gemm_batched_f32((const float* const*)A, (const float* const*)B, (float* const*)C, m, n, k, b, ta, tb, alpha, beta, nullptr);
}
inline int gemm_call(double** A, double** B, double** C, std::size_t m, std::size_t n, std::size_t k, std::size_t b, int ta,
                     int tb, double alpha, double beta) {
  return sm_gemm_batched_f64((const double* const*)A, (const double* const*)B, (double* const*)C, m, n, k, b, ta, tb, alpha, beta, nullptr);
}
}  // namespace detail

template <typename type_t>
float gemm(type_t** A_ptrs,
           type_t** B_ptrs,
           type_t** C_ptrs,
           std::size_t m,
           std::size_t n,
           std::size_t k,
           std::size_t batch_size,
           operation_t transpose_a = operation_t::N,
           operation_t transpose_b = operation_t::N,
           type_t alpha = (type_t)1.0f,
           type_t beta = (type_t)0.0f) {
  util::range_t range("batched-GEMM");
  util::timer_t timer;
  timer.begin();
  const int rc = detail::gemm_call(A_ptrs, B_ptrs, C_ptrs, m, n, k, batch_size, (int)transpose_a, (int)transpose_b, alpha, beta);
  // reference behaviour (gemm.hxx:82-87): a failing status is printed, not propagated
  if (rc != SM_STATUS_SUCCESS)
    std::cout << "error: sm_gemm_batched exited with an error: " << rc << " (" << sm_last_error() << ")" << std::endl;
  return timer.end();
}
}  // namespace batched
}  // namespace sparsifyme
