// launch.hxx -- per-batch launch state.  The reference's launch_t (include/sparsify.me/util/launch.hxx:19-42)
// carries a stream, an event, a cuSPARSE handle and a workspace; the MI355X kernels need no vendor
// handle and no workspace, so `handle` is kept only as an opaque slot for source compatibility.
// One batched kernel launch replaces the reference's one-host-thread-per-batch scheme
// (spmm.hxx:94), so these configs are not used by the operators themselves.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

#include <sparsify.me/util/timer.hxx>

namespace sparsifyme {
namespace util {

struct launch_t {
  hipStream_t stream = nullptr;
  hipEvent_t event = nullptr;
  void* handle = nullptr;  // no vendor library handle exists in this build
  void* buffer = nullptr;
  std::size_t buffer_size = 0;
};

inline void create_launch_configs(std::vector<launch_t>& configs) {
  for (auto& c : configs) (void)hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
}

inline void destroy_launch_configs(std::vector<launch_t>& configs) {
  for (auto& c : configs) {
    if (c.buffer) (void)hipFree(c.buffer);
    c.buffer = nullptr;
    c.buffer_size = 0;
    if (c.stream) (void)hipStreamDestroy(c.stream);  // the reference leaks its streams; this build does not
    c.stream = nullptr;
  }
}

}  // namespace util
}  // namespace sparsifyme
