// Microbenchmark (diagnostic tool, round 3): how much HBM bandwidth ONE CU can pull when fewer than all 256 CUs are
// streaming (the few-tile matmul layers run 196 / 98 workgroups, one per CU), as a function of the bytes it keeps in
// flight: `blocks` workgroups (one per CU: LDS sized so that two do not fit), `waves` waves each, every wave keeps
// `depth` x 1 KiB plain 16-byte loads (non-temporal) in flight over a private stream.
// build: hipcc --offload-arch=gfx950 -O3 tools/fillrate2.hip -o tools/fillrate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int DEPTH>
__global__ void stream(const char* __restrict__ src, size_t per_wave, unsigned* sink) {
  extern __shared__ char lds[];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* p = src + ((size_t)blockIdx.x * nw + wave) * per_wave + lane * 16u;
  const int iters = (int)(per_wave / (DEPTH * 1024));
  u4 acc = {0, 0, 0, 0};
  u4 r[DEPTH];
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) r[j] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p + j * 1024));
  for (int it = 1; it < iters; ++it) {
    p += DEPTH * 1024;
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      acc ^= r[j];  // waits for the oldest load only (counted vmcnt), then re-issues its slot
      r[j] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p + j * 1024));
    }
  }
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) acc ^= r[j];
  if (acc[0] == 0x12345678u && acc[1] == 1u) sink[0] = acc[2] + acc[3];
  if (lds[0] == 77 && per_wave == 1) sink[1] = 1;
}

template <int DEPTH>
static void run(const char* d, int blocks, int waves, size_t total, unsigned* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t per_wave = total / ((size_t)blocks * waves);
  per_wave -= per_wave % (DEPTH * 1024);
  const size_t lds = 96 * 1024;  // one workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  stream<DEPTH><<<blocks, waves * 64, lds>>>(d, per_wave, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  stream<DEPTH><<<blocks, waves * 64, lds>>>(d, per_wave, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)per_wave * blocks * waves;
  const double g = bytes / (ms * 1e-3) / 1e9;
  printf("%6d %5d %5d %9d %10.0f %10.1f %8.1f\n", blocks, waves, DEPTH, waves * DEPTH, g, g / blocks, ms * 1e3);
}

int main() {
  const size_t big = (size_t)1 << 30;
  char* d; unsigned* sink;
  CK(hipMalloc(&d, big)); CK(hipMemset(d, 1, big)); CK(hipMalloc(&sink, 64));
  printf("%6s %5s %5s %9s %10s %10s %8s\n", "blocks", "waves", "depth", "KiB/CU", "GB/s", "GB/s/CU", "us");
  for (int blocks : {32, 64, 98, 128, 196, 256})
    for (int waves : {4, 8, 16}) {
      run<2>(d, blocks, waves, big, sink);
      run<4>(d, blocks, waves, big, sink);
      run<8>(d, blocks, waves, big, sink);
      run<16>(d, blocks, waves, big, sink);
    }
  return 0;
}
