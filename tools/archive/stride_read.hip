// Microbenchmark (diagnostic tool): HBM read rate of the fused kernels' A access pattern -- a workgroup walks a panel
// of ROWS rows along k, each wave-load covering (1024 / RUN) rows x RUN contiguous bytes at a row pitch of `pitch`
// bytes -- for RUN = 128 (what a 64-deep stage gives), 256, 512, 1024, against a fully contiguous stream.
// build: hipcc --offload-arch=gfx950 -O3 tools/stride_read.hip -o tools/stride_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// panel: 128 rows per workgroup (4 waves x 32 rows), pitch bytes per row, kbytes per row in total
template <int RUN>
__global__ __launch_bounds__(256) void walk(const char* __restrict__ A, size_t pitch, size_t kbytes, unsigned* sink, char* W = nullptr) {
  constexpr int RPL = 1024 / RUN;          // rows per wave-load
  constexpr int LPS = 32 / RPL;            // loads per wave to cover its 32 rows x RUN bytes
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const size_t row0 = (size_t)blockIdx.x * 128 + wave * 32;
  const unsigned rl = lane / (RUN / 16), cb = (lane % (RUN / 16)) * 16;
  u4 acc = {0, 0, 0, 0};
  for (size_t k0 = 0; k0 < kbytes; k0 += RUN) {
    u4 v[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i)
      v[i] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(A + (row0 + i * RPL + rl) * pitch + k0 + cb));
#pragma unroll
    for (int i = 0; i < LPS; ++i) acc ^= v[i];
  }
  if (acc[0] == 0x12345678u && acc[1] == 1u) sink[0] = acc[2] + acc[3];
  // optional write stream: 128 B per row (the C tile of an n = 64 layer), 16 bytes per lane, 8 rows per wave-store
  if (W) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u4*>(W + (row0 + i * 8 + (lane >> 3)) * 128 + (lane & 7u) * 16) = acc;
  }
}

template <int RUN>
static void run(const char* d, size_t rows, size_t pitch, unsigned* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(rows / 128);
  walk<RUN><<<grid, 256>>>(d, pitch, pitch, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  walk<RUN><<<grid, 256>>>(d, pitch, pitch, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("run %4d B per row per step, pitch %5zu B, %7zu rows: %7.3f ms  %7.1f GB/s\n", RUN, pitch, rows, ms, rows * pitch / (ms * 1e-3) / 1e9);
}

static void run_rw(const char* d, size_t rows, size_t pitch, unsigned* sink, char* w) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(rows / 128);
  walk<128><<<grid, 256>>>(d, pitch, pitch, sink, w);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  walk<128><<<grid, 256>>>(d, pitch, pitch, sink, w);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("run  128 B + 128 B written per row, pitch %5zu B, %7zu rows: %7.3f ms  %7.1f GB/s read+write\n", pitch, rows, ms, rows * (pitch + 128) / (ms * 1e-3) / 1e9);
}

int main() {
  unsigned* sink; CK(hipMalloc(&sink, 64));
  const size_t pitches[3] = {1152, 512, 4608};
  for (size_t pitch : pitches) {
    const size_t rows = ((size_t)1 << 30) / pitch / 128 * 128;  // ~1 GiB
    char* d; CK(hipMalloc(&d, rows * pitch)); CK(hipMemset(d, 1, rows * pitch));
    run<128>(d, rows, pitch, sink);
    { char* w; CK(hipMalloc(&w, rows * 128)); run_rw(d, rows, pitch, sink, w); CK(hipFree(w)); }
    run<256>(d, rows, pitch, sink);
    run<512>(d, rows, pitch, sink);
    if (pitch % 1024 == 0) run<1024>(d, rows, pitch, sink);
    CK(hipFree(d));
  }
  return 0;
}
