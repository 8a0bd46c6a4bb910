// pending.hip -- entry points declared in include/sparsifyme.h whose kernels are not written yet.
// They fail loudly (SM_STATUS_NOT_SUPPORTED + message); nothing here computes on the host.
#include "sm_common.h"
using namespace sm;
#define SM_PENDING(NAME) \
  set_error(NAME ": kernel not implemented in this build"); \
  return SM_STATUS_NOT_SUPPORTED
extern "C" {
int sm_spmma_f32(const void*, const float*, float*, size_t, size_t, size_t, size_t, size_t, size_t, float, float, sm_stream_t) { SM_PENDING("sm_spmma_f32"); }
int sm_gemm_batched_f32(const float* const*, const float* const*, float* const*, size_t, size_t, size_t, size_t, int, int, float, float, sm_stream_t) { SM_PENDING("sm_gemm_batched_f32"); }
int sm_gemm_batched_f64(const double* const*, const double* const*, double* const*, size_t, size_t, size_t, size_t, int, int, double, double, sm_stream_t) { SM_PENDING("sm_gemm_batched_f64"); }
int sm_gemm_rowmajor_f32(const float*, const float*, float*, size_t, size_t, size_t, size_t, size_t, size_t, size_t, size_t, float, float, sm_stream_t) { SM_PENDING("sm_gemm_rowmajor_f32"); }
int sm_spmm_bell_f32(const float*, const uint64_t*, size_t, size_t, size_t, size_t, const float*, float*, size_t, float, float, sm_stream_t) { SM_PENDING("sm_spmm_bell_f32"); }
int sm_spmm_coo_f32(size_t, size_t, size_t, size_t, size_t, const int*, const int*, const float*, const float*, float*, float, float, sm_stream_t) { SM_PENDING("sm_spmm_coo_f32"); }
}
