// Shared host/device helpers for libsparsifyme.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sparsifyme.h"

namespace sm {

typedef _Float16 half_t;
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// Records the message sm_last_error() returns (thread local, defined in api.hip).
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SM_STATUS_LAUNCH_FAILED;
  }
  return SM_STATUS_SUCCESS;
}

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline size_t ceil_div(size_t x, size_t y) { return (x + y - 1) / y; }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Layout of the 2:4 compressed blob (include/sparsifyme.h header comment).
struct BlobLayout {
  size_t kc, M, meta_off, total;
};
inline BlobLayout blob_layout(size_t m, size_t k, size_t elt, size_t batch) {
  BlobLayout L;
  L.kc = round_up(k, 64);
  L.M = m * batch;
  L.meta_off = round_up(L.M * (L.kc / 2) * elt, 256);
  L.total = L.meta_off + round_up(L.M * (L.kc / 8), 256);
  return L;
}

// Grid size for grid-stride streaming kernels: enough blocks to fill 256 CUs x 8, capped.
inline unsigned stream_grid(size_t work_items, unsigned block) {
  size_t g = ceil_div(work_items, block);
  if (g > 256u * 16u) g = 256u * 16u;
  if (g == 0) g = 1;
  return (unsigned)g;
}

}  // namespace sm
