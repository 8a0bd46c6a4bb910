// gen.hxx -- device-side uniform fill (reference: include/sparsify.me/util/gen.hxx:8-21, a Thrust
// transform over a default-seeded engine).  Here: the counter-based fill kernel of
// libsparsifyme.so (sm_fill_uniform_*): element i depends only on (seed, i).
#pragma once
#include <cstdint>

#include <sparsifyme.h>
#include <sparsify.me/containers/vector.hxx>
#include <sparsify.me/util/util.hxx>

namespace sparsifyme {
namespace util {
namespace random {

inline void uniform_distribution(device_vector<float>& input, float begin = 0.0f, float end = 1.0f, std::uint64_t seed = 0) {
  (void)sm_fill_uniform_f32(input.data().get(), input.size(), seed, begin, end, nullptr);
  (void)hipStreamSynchronize(nullptr);
}
inline void uniform_distribution(device_vector<_Float16>& input, float begin = 0.0f, float end = 1.0f, std::uint64_t seed = 0) {
  (void)sm_fill_uniform_f16(input.data().get(), input.size(), seed, begin, end, nullptr);
  (void)hipStreamSynchronize(nullptr);
}
template <typename T>
inline void uniform_distribution(host_vector<T>& input, T begin = T(0), T end = T(1)) {
  for (auto& x : input) x = get_random<T>(begin, end);
}

}  // namespace random
}  // namespace util
}  // namespace sparsifyme
