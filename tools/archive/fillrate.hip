// Microbenchmark (diagnostic tool): per-CU fill rate of the two global->on-chip paths on gfx950,
//   (a) global_load_lds_dwordx4 (LDS-DMA)      (b) global_load_dwordx4 to VGPRs,
// from an L2-resident buffer (1 MiB, re-read) and from HBM (2 GiB stream), for 1/2/4/8/16 waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/fillrate.hip -o tools/fillrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// each wave: `iters` rounds of 8 x 1 KiB loads; addresses walk `span` bytes (power of two) per block
template <bool DMA>
__global__ void fill(const char* __restrict__ src, size_t span, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + ((size_t)blockIdx.x * span) ;
  size_t off = ((size_t)wave * 8192u) & (span - 1);
  u4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const char* p = base + ((off + j * 1024u) & (span - 1)) + lane * 16u;
      if constexpr (DMA) {
        __builtin_amdgcn_global_load_lds((gptr_t*)p, (lptr_t*)(lds + wave * 8192u + j * 1024u), 16, 0, 0);
      } else {
        const u4 v = *reinterpret_cast<const u4*>(p);
        acc ^= v;
      }
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    off = (off + (size_t)nw * 8192u) & (span - 1);
  }
  if (acc[0] == 0x12345678u && acc[1] == 1u) sink[0] = acc[2] + acc[3];
  if constexpr (DMA) { if (lds[threadIdx.x] == 77 && iters < 0) sink[1] = 1; }
}

template <bool DMA>
static double run(const char* d, size_t span, int blocks, int waves, int iters, unsigned* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = DMA ? (size_t)waves * 8192 : 0;
  fill<DMA><<<blocks, waves * 64, lds>>>(d, span, iters, sink);  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  fill<DMA><<<blocks, waves * 64, lds>>>(d, span, iters, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)blocks * waves * iters * 8192.0;
  return bytes / (ms * 1e-3) / 1e9;  // GB/s total
}

int main() {
  const size_t big = (size_t)2 << 30;
  char* d; unsigned* sink;
  CK(hipMalloc(&d, big)); CK(hipMemset(d, 1, big)); CK(hipMalloc(&sink, 64));
  printf("%-6s %-5s %5s %12s %12s\n", "src", "path", "waves", "GB/s total", "GB/s per CU");
  for (int hbm = 0; hbm < 2; ++hbm)
    for (int dma = 0; dma < 2; ++dma)
      for (int waves : {1, 2, 4, 8, 16}) {
        const int blocks = 256;
        // L2 case: every block re-reads its own 64 KiB (stays in L2); HBM case: 8 MiB per block, streamed once
        const size_t span = hbm ? ((size_t)8 << 20) : ((size_t)64 << 10);
        const int iters = hbm ? (int)(span / ((size_t)waves * 8192)) : 2000 / waves + 64;
        const double g = dma ? run<true>(d, span, blocks, waves, iters, sink) : run<false>(d, span, blocks, waves, iters, sink);
        printf("%-6s %-5s %5d %12.0f %12.1f\n", hbm ? "HBM" : "L2", dma ? "dma" : "vgpr", waves, g, g / 256.0);
      }
  return 0;
}
