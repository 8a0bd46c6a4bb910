// Shared host/device helpers for libsparsifyme.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>

#include "../../include/sparsifyme.h"

namespace sm {

typedef _Float16 half_t;
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
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

// One-shot opt-in of a kernel to more than 64 KiB of dynamic LDS, done once PER DEVICE (the attribute belongs to the
// function object of the current device; one bit per device ordinal, set after the call succeeded, so two threads
// racing on the first launch at worst both set it).  A refused opt-in is reported here, not as a later launch error.
struct LdsOptIn {
  std::atomic<unsigned long long> done[4];  // 256 device ordinals
};
int ensure_dyn_lds(LdsOptIn& s, const void* fn, size_t bytes, const char* what);
// Compute units of the current device, queried once per device ordinal (hipGetDeviceProperties costs tens of
// microseconds; launchers that size a grid by the CU count call this on every launch).
int device_cu_count();
// Tuning hooks of tools/ (SM_* environment variables) exist only in -DSM_TUNING builds; the product library
// never reads the environment.
#ifdef SM_TUNING
inline const char* tuning_env(const char* name) { return getenv(name); }
#else
inline const char* tuning_env(const char*) { return nullptr; }
#endif
inline int tuning_int(const char* name, int dflt) {
  const char* v = tuning_env(name);
  return v ? atoi(v) : dflt;
}

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline size_t ceil_div(size_t x, size_t y) { return (x + y - 1) / y; }
// Raise a device flag from one lane of a wave: the relaxed read first, so that once the flag is up the thousands of waves that also
// found something do not queue their atomics on one address (an all-invalid 462 MB operand spent 100 us of a 214 us check pass there).
#ifdef __HIPCC__
__device__ __forceinline__ void raise_flag(int* flag) {
  if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
}
#endif
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
// `uncapped`: one item per thread and no second trip (round 5: the copy probe's finding -- workgroups handed out in address order keep the chip's traffic
// inside a few megabytes at a time; a capped grid's stride loop smears it.  It pays where the kernel is a plain copy-like stream: STRIP prune of the 462 MB
// operand 176 -> 155 us, 231 MB 85 -> 80; the check, compress, TILE and one-pass kernels do not move, profiles/prune_ab_r05as.txt).
inline unsigned stream_grid(size_t work_items, unsigned block, bool uncapped = false) {
  size_t g = ceil_div(work_items, block);
  const size_t cap = (size_t)tuning_int("SM_STREAM_GRID_CAP", uncapped ? 0 : 256 * 16);  // tuning aid (A/B of the cap): 0 = none
  if (cap && g > cap) g = cap;
  if (g > 0x7fffffffull) g = 0x7fffffffull;
  if (g == 0) g = 1;
  return (unsigned)g;
}

}  // namespace sm
