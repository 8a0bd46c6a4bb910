// trace.hxx -- named ranges for profilers, the role NVTX plays in the reference (nvtxRangePushA("batched-SpMM") /
// nvtxRangePop() around the batch loop, include/sparsify.me/spmm.hxx:92,121).  Here the ranges are ROCTX ranges
// (shown by `rocprofv3 --marker-trace`), compiled in with -DSPARSIFYME_ROCTX (link -lroctx64) and free otherwise.
#pragma once
#ifdef SPARSIFYME_ROCTX
#include <roctracer/roctx.h>
#endif

namespace sparsifyme {
namespace util {

// RAII: the range covers the scope, whichever way it is left.
struct range_t {
  explicit range_t(const char* name) {
#ifdef SPARSIFYME_ROCTX
    roctxRangePushA(name);
#else
    (void)name;
#endif
  }
  ~range_t() {
#ifdef SPARSIFYME_ROCTX
    roctxRangePop();
#endif
  }
  range_t(const range_t&) = delete;
  range_t& operator=(const range_t&) = delete;
};

}  // namespace util
}  // namespace sparsifyme
