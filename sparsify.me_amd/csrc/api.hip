// api.hip -- library-level entry points of libsparsifyme.so: version, device check, error text,
// and the counter-based uniform fill (support kernel K10 of SURVEY.md 2.2; replaces the Thrust
// transform of include/sparsify.me/util/gen.hxx:12-20).
#include <stdarg.h>
#include <string.h>

#include "sm_common.h"

namespace sm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int ensure_dyn_lds(LdsOptIn& s, const void* fn, size_t bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) {
    (void)hipGetLastError();
    set_error("%s: cannot identify the current device", what);
    return SM_STATUS_LAUNCH_FAILED;
  }
  const unsigned long long bit = 1ull << (dev & 63);
  std::atomic<unsigned long long>& w = s.done[dev >> 6];
  if (w.load(std::memory_order_acquire) & bit) return SM_STATUS_SUCCESS;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: device %d refused %zu bytes of dynamic LDS: %s", what, dev, bytes, hipGetErrorString(e));
    return SM_STATUS_LAUNCH_FAILED;
  }
  w.fetch_or(bit, std::memory_order_release);
  return SM_STATUS_SUCCESS;
}

int device_cu_count() {
  static std::atomic<int> cus[256];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) {
    (void)hipGetLastError();
    return 256;
  }
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c > 0) return c;
  if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  cus[dev].store(c, std::memory_order_relaxed);
  return c;
}

// splitmix64 of (seed, i): element i depends on nothing else, so a fill is reproducible for any
// grid shape and can be regenerated per GPU from (seed, layer, batch) without host traffic.
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t i) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * 5.9604644775390625e-8f;  // 24 bits -> [0, 1)
}

template <typename T>
__global__ __launch_bounds__(256) void fill_uniform_kernel(T* out, size_t count, uint64_t seed, float lo, float span) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const float v = __fadd_rn(lo, __fmul_rn(span, uniform01(seed, i)));
    out[i] = (T)v;
  }
}

template <typename T>
static int launch_fill(void* out, size_t count, uint64_t seed, float lo, float hi, hipStream_t st) {
  if (!out && count) {
    set_error("sm_fill_uniform: null output");
    return SM_STATUS_INVALID_VALUE;
  }
  if (count == 0) return SM_STATUS_SUCCESS;
  fill_uniform_kernel<T><<<stream_grid(count, 256), 256, 0, st>>>((T*)out, count, seed, lo, hi - lo);
  return check_launch("fill_uniform_kernel");
}

// Plain streaming copy: the yardstick bench.py times in the same process as the step -- what HBM delivers to the simplest possible kernel on this
// part, on this day, on this data.  ONE 16-byte non-temporal load and store per thread, no loop (round 5; tools/probes/copy_probe.hip,
// profiles/copy_probe_r05a{p,q,r}.txt, pseudo-random data, 3.7 GB each way): 6.5-6.6 TB/s where the grid-stride form it replaces (4 096 workgroups, four
// pieces per thread and trip) gave 5.3-5.9 -- the dispatcher handing out workgroups in address order keeps the chip's reads and writes inside a few
// megabytes at any moment, a grid-stride loop smears them over gridDim x 16 KiB; one contiguous chunk per workgroup is worst (4.4-4.8).  The rate also
// depends on the DATA: the same copy of a buffer of equal bytes runs 15 % faster (6.4 against 5.4 TB/s) -- which is why this yardstick copies the step's
// own operands' worth of random halves, and why the guide's 6.29 TB/s float4 copy is reported beside it rather than instead of it.
__global__ __launch_bounds__(256) void copy_bytes_kernel(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace sm

extern "C" {

int sm_copy_bytes(const void* src, void* dst, size_t bytes, sm_stream_t s) {
  if (bytes == 0) return SM_STATUS_SUCCESS;
  if (!src || !dst || (bytes & 15u) || !sm::aligned16(src) || !sm::aligned16(dst)) {
    sm::set_error("sm_copy_bytes: needs non-null 16-byte aligned buffers and a byte count that is a multiple of 16");
    return SM_STATUS_INVALID_VALUE;
  }
  const size_t n16 = bytes / 16;
  const size_t blocks = (n16 + 255) / 256;
  if (blocks > 0x7fffffffull) {
    sm::set_error("sm_copy_bytes: more than 2^31 workgroups (copy in pieces)");
    return SM_STATUS_NOT_SUPPORTED;
  }
  sm::copy_bytes_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)s>>>((const sm::u4*)src, (sm::u4*)dst, n16);
  return sm::check_launch("copy_bytes_kernel");
}

const char* sm_version(void) { return "sparsifyme-amd 0.1.0 (gfx950)"; }

const char* sm_last_error(void) { return sm::g_err; }

int sm_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    (void)hipGetLastError();
    sm::set_error("no HIP device visible");
    return SM_STATUS_NO_DEVICE;
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    sm::set_error("cannot query the current HIP device");
    return SM_STATUS_NO_DEVICE;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    sm::set_error("device %d is %s; this library contains gfx950 code only", dev, prop.gcnArchName);
    return SM_STATUS_NO_DEVICE;
  }
  return SM_STATUS_SUCCESS;
}

int sm_fill_uniform_f16(void* out, size_t count, uint64_t seed, float lo, float hi, sm_stream_t s) {
  return sm::launch_fill<_Float16>(out, count, seed, lo, hi, (hipStream_t)s);
}
int sm_fill_uniform_bf16(void* out, size_t count, uint64_t seed, float lo, float hi, sm_stream_t s) {
  return sm::launch_fill<__bf16>(out, count, seed, lo, hi, (hipStream_t)s);
}
int sm_fill_uniform_f32(float* out, size_t count, uint64_t seed, float lo, float hi, sm_stream_t s) {
  return sm::launch_fill<float>(out, count, seed, lo, hi, (hipStream_t)s);
}

}  // extern "C"
