// copy_probe.hip -- what a device copy can reach on this part: variants of the 16-byte streaming copy that bench.py uses as its yardstick
// (sm_copy_bytes: 5.4-5.7 TB/s by box) against the guide's 6.29 TB/s float4 copy.  hipcc --offload-arch=gfx950 -O3 -o tools/probes/copy_probe tools/probes/copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_stride(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n16; i += stride) {
    u4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (i + (size_t)j * 256 < n16) v[j] = NTL ? __builtin_nontemporal_load(src + i + (size_t)j * 256) : src[i + (size_t)j * 256];
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (i + (size_t)j * 256 < n16) { if (NTS) __builtin_nontemporal_store(v[j], dst + i + (size_t)j * 256); else dst[i + (size_t)j * 256] = v[j]; }
  }
}
// one block = one contiguous chunk (no grid stride): n16 / gridDim elements each, U loads in flight
template <int U, int T>
__global__ __launch_bounds__(T) void copy_chunk(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x, b0 = per * blockIdx.x, b1 = b0 + per < n16 ? b0 + per : n16;
  for (size_t i = b0 + threadIdx.x; i < b1; i += (size_t)T * U) {
    u4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (i + (size_t)j * T < b1) v[j] = __builtin_nontemporal_load(src + i + (size_t)j * T);
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (i + (size_t)j * T < b1) __builtin_nontemporal_store(v[j], dst + i + (size_t)j * T);
  }
}
// through LDS: LDS-DMA in (no registers), ds_read, store; ring of R slots of 4 KiB per wave
template <int R>
__global__ __launch_bounds__(256) void copy_dma(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  char* ring = smem + wave * (R * 1024);
  const size_t nwaves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + wave;
  const size_t units = n16 / 64;  // 1 KiB units
  size_t u = w;
  int issued = 0;
  auto issue = [&](size_t uu, int slot) { __builtin_amdgcn_global_load_lds((gptr_t*)(src + uu * 64 + lane), (lptr_t*)(ring + slot * 1024), 16, 0, 2); };
#pragma unroll
  for (int s = 0; s < R - 1; ++s) { if (u + (size_t)s * nwaves < units) issue(u + (size_t)s * nwaves, s); ++issued; }
  int slot = 0;
  for (; u < units; u += nwaves) {
    const size_t un = u + (size_t)(R - 1) * nwaves;
    if (un < units) issue(un, (slot + R - 1) % R); else asm volatile("" ::: "memory");
    if (un < units) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u4 v = *reinterpret_cast<const u4*>(ring + slot * 1024 + lane * 16);
    __builtin_nontemporal_store(v, dst + u * 64 + lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    slot = (slot + 1) % R;
  }
}
// the direct fused kernel's traffic without its arithmetic: a block = one tile of 128 rows x K halves of A (rows back to back: one contiguous
// block of 128 * K * 2 bytes), read either as the kernel reads it -- K / 64 passes over the 128 rows, 128 bytes of each row per pass -- or in
// address order; then 128 x NB bytes of C written.  XOR-reduced into the C tile so that nothing is optimised away.
template <bool SEQ>
__global__ __launch_bounds__(256) void tile_pattern(const u4* __restrict__ A, u4* __restrict__ C, int kbytes, int cbytes_per_row) {
  const size_t tile = blockIdx.x;
  const char* a0 = reinterpret_cast<const char*>(A) + tile * (size_t)128 * kbytes;
  u4 acc = {0u, 0u, 0u, 0u};
  const int nst = kbytes / 128;
  if (SEQ) {
    const int n16 = 128 * kbytes / 16;
    for (int i = threadIdx.x; i < n16; i += 1024) {
      u4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = i + 256 * j < n16 ? __builtin_nontemporal_load(reinterpret_cast<const u4*>(a0) + i + 256 * j) : u4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
  } else {
    for (int s = 0; s < nst; ++s) {
      u4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // 8 lanes per row piece of 128 bytes, 32 rows per pass of the block
        const int row = (threadIdx.x >> 3) + 32 * j, c = threadIdx.x & 7;
        v[j] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(a0 + (size_t)row * kbytes + s * 128 + c * 16));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
  }
  u4* c0 = reinterpret_cast<u4*>(reinterpret_cast<char*>(C) + tile * (size_t)128 * cbytes_per_row);
  for (int i = threadIdx.x; i < 128 * cbytes_per_row / 16; i += 256) __builtin_nontemporal_store(acc + (unsigned)i, c0 + i);
}
__global__ void fill_random(u4* p, size_t n16, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    u4 v;
    for (int e = 0; e < 4; ++e) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v[e] = x; }
    p[i] = v;
  }
}
int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)3718053888ull) / 16 * 16;
  u4 *src, *dst;
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const size_t n16 = bytes / 16;
  const bool random = argc > 2 && atoi(argv[2]) != 0;
  hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
  if (random) { fill_random<<<4096, 256>>>(src, n16, 12345u); fill_random<<<4096, 256>>>(dst, n16, 999u); }
  printf("# %zu bytes each way, source data: %s\n", bytes, random ? "pseudo-random words" : "every byte 0x01");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    printf("%-52s %8.3f ms  %7.1f GB/s read+write%s\n", name, best, 2.0 * bytes / best / 1e6, err == hipSuccess ? "" : "  (ERROR)");
    fflush(stdout);
  };
  for (unsigned grid : {2048u, 4096u, 8192u, 16384u, 65536u}) {
    char nm[96];
    snprintf(nm, 96, "stride U=4 nt/nt grid %u", grid); run(nm, [&] { copy_stride<4, true, true><<<grid, 256>>>(src, dst, n16); });
  }
  run("stride U=4 plain/plain grid 4096", [&] { copy_stride<4, false, false><<<4096, 256>>>(src, dst, n16); });
  run("stride U=4 nt loads, plain stores", [&] { copy_stride<4, true, false><<<4096, 256>>>(src, dst, n16); });
  run("stride U=4 plain loads, nt stores", [&] { copy_stride<4, false, true><<<4096, 256>>>(src, dst, n16); });
  run("stride U=1 nt/nt grid 16384", [&] { copy_stride<1, true, true><<<16384, 256>>>(src, dst, n16); });
  run("stride U=2 nt/nt grid 8192", [&] { copy_stride<2, true, true><<<8192, 256>>>(src, dst, n16); });
  run("stride U=8 nt/nt grid 2048", [&] { copy_stride<8, true, true><<<2048, 256>>>(src, dst, n16); });
  run("stride U=8 nt/nt grid 4096", [&] { copy_stride<8, true, true><<<4096, 256>>>(src, dst, n16); });
  run("stride U=16 nt/nt grid 2048", [&] { copy_stride<16, true, true><<<2048, 256>>>(src, dst, n16); });
  run("chunk U=4 T=256 grid 2048", [&] { copy_chunk<4, 256><<<2048, 256>>>(src, dst, n16); });
  run("chunk U=4 T=256 grid 8192", [&] { copy_chunk<4, 256><<<8192, 256>>>(src, dst, n16); });
  run("chunk U=4 T=1024 grid 1024", [&] { copy_chunk<4, 1024><<<1024, 1024>>>(src, dst, n16); });
  run("chunk U=2 T=1024 grid 2048", [&] { copy_chunk<2, 1024><<<2048, 1024>>>(src, dst, n16); });
  run("one element per thread (no loop)", [&] { copy_stride<1, true, true><<<(unsigned)((n16 + 255) / 256), 256>>>(src, dst, n16); });
  run("one element per thread, plain", [&] { copy_stride<1, false, false><<<(unsigned)((n16 + 255) / 256), 256>>>(src, dst, n16); });
  run("LDS-DMA ring 4, grid 2048", [&] { copy_dma<4><<<2048, 256, 4 * 4 * 1024>>>(src, dst, n16); });
  run("LDS-DMA ring 8, grid 2048", [&] { copy_dma<8><<<2048, 256, 4 * 8 * 1024>>>(src, dst, n16); });
  run("LDS-DMA ring 8, grid 1024", [&] { copy_dma<8><<<1024, 256, 4 * 8 * 1024>>>(src, dst, n16); });
  for (int kb : {1152, 512, 2304}) {
    const size_t tiles = bytes / ((size_t)128 * kb) < 9408 * 2 ? bytes / ((size_t)128 * kb) : 9408 * 2;
    const double mb = tiles * (128.0 * kb + 128.0 * 128) / 1e6;
    char nm[96];
    auto runp = [&](const char* name, auto launch) {
      float best = 1e9f;
      for (int r = 0; r < 5; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms; }
      printf("%-52s %8.3f ms  %7.1f GB/s (A read + C written, %0.0f MB)\n", name, best, mb / best / 1e3 * 1e3 / 1e3 * 1e3, mb); fflush(stdout);
    };
    snprintf(nm, 96, "tile pattern, row %d B: as the kernel reads", kb); runp(nm, [&] { tile_pattern<false><<<(unsigned)tiles, 256>>>(src, dst, kb, 128); });
    for (int lds_kb : {24, 32, 48, 64, 80}) {  // workgroups per CU limited the way the kernel's LDS limits them (160 KiB per CU)
      hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_pattern<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      snprintf(nm, 96, "  same, %d KiB of LDS per workgroup (%d per CU)", lds_kb, 160 / lds_kb);
      runp(nm, [&] { tile_pattern<false><<<(unsigned)tiles, 256, (size_t)lds_kb * 1024>>>(src, dst, kb, 128); });
    }
    snprintf(nm, 96, "tile pattern, row %d B: in address order", kb); runp(nm, [&] { tile_pattern<true><<<(unsigned)tiles, 256>>>(src, dst, kb, 128); });
  }
  run("hipMemcpyAsync D2D", [&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
