// timer.hxx -- HIP-event stopwatch with the reference's semantics (include/sparsify.me/util/timer.hxx:24-55):
// begin(stream) records the start event and waits for it; end(stream) records the stop event,
// waits for it and returns the elapsed milliseconds, so an operator that ends with timer.end()
// is blocking, exactly as the reference's operators are.
#pragma once
#include <hip/hip_runtime.h>

namespace sparsifyme {
namespace util {

struct timer_t {
  float time = 0.0f;

  timer_t() {
    (void)hipEventCreate(&start_);
    (void)hipEventCreate(&stop_);
  }
  ~timer_t() {
    (void)hipEventDestroy(start_);
    (void)hipEventDestroy(stop_);
  }
  timer_t(const timer_t&) = delete;
  timer_t& operator=(const timer_t&) = delete;

  void begin(hipStream_t stream = 0) {
    (void)hipEventRecord(start_, stream);
    (void)hipEventSynchronize(start_);
  }
  float end(hipStream_t stream = 0) {
    (void)hipEventRecord(stop_, stream);
    (void)hipEventSynchronize(stop_);
    (void)hipEventElapsedTime(&time, start_, stop_);
    return milliseconds();
  }
  float seconds() { return time * 1e-3f; }
  float milliseconds() { return time; }

 private:
  hipEvent_t start_, stop_;
};

}  // namespace util
}  // namespace sparsifyme
