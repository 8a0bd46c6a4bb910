// sparsify.hxx -- sparsifyme::sparsify: positional block pruning with a size_t mask.
// Same signature and effect as the reference's include/sparsify.me/sparsify.hxx:24-82 (there a
// Thrust fill + a device lambda; here one fused HIP kernel behind sm_sparsify_positional):
// mask[0..m*n) = 1; in every run of BLK_M*BLK_N consecutive elements the first
// floor(BLK_M*BLK_N*sparsity_factor) offsets in the reference's visit order (h + w*BLK_N, h-major:
// 0,2,1,3 for 2x2) are zeroed in `weights` and `mask`.  In place on device pointers; asynchronous
// on `stream`, as the reference's is.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <iostream>

#include <sparsifyme.h>
#include <sparsify.me/util/util.hxx>

namespace sparsifyme {
template <std::size_t BLK_M = 2, std::size_t BLK_N = 2, typename type_t>
void sparsify(type_t* weights,
              std::size_t* mask,
              std::size_t const& m,
              std::size_t const& n,
              float sparsity_factor = 0.5,
              hipStream_t stream = 0) {
  static_assert(sizeof(std::size_t) == sizeof(std::uint64_t), "the mask is 64 bits per element");
  const int rc = sm_sparsify_positional(weights, reinterpret_cast<std::uint64_t*>(mask), m, n, sizeof(type_t),
                                        BLK_M, BLK_N, sparsity_factor, stream);
  // the reference never reports failure from this operator; keep that, but say what went wrong
  if (rc != SM_STATUS_SUCCESS) std::cerr << "sparsifyme::sparsify: " << sm_last_error() << std::endl;
}
}  // namespace sparsifyme
