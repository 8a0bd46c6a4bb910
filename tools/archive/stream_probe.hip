// Microbenchmark (diagnostic tool, round 3): could a BARRIER-FREE direct kernel stream A faster than the barrier-per-stage
// one?  Skeleton of that design without the arithmetic: one workgroup of NWAVES waves per CU (LDS reserved so that only one
// fits, as a resident B tile would force), every wave walks its own 16 (or 32) rows along k: each 64-k stage is one (two)
// 1 KiB LDS-DMA pieces (8 rows x 128 B at a row pitch) into a wave-private ring of DEPTH stages, a counted vmcnt wait for the
// oldest stage, two ds_read_b128 per lane of it, VALU dummy instructions per stage, and at the end of a row block a 16 x 64
// fp16 C piece written (nt).  No s_barrier after the start.  Reports TB/s of A bytes.
// build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int DEPTH, int VALU, int RPW /*rows per wave: 16 or 32*/>
__global__ __launch_bounds__(1024) void stream(const char* __restrict__ A, size_t pitch, int nkt, size_t rows, unsigned lds_reserve, char* __restrict__ C,
                                               unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PCS = RPW / 8;                 // 1 KiB pieces per stage
  constexpr int STG = PCS * 1024;
  const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  char* ring = smem + lds_reserve + wave * (DEPTH * STG);
  const size_t blocks = rows / RPW;            // row blocks in all
  unsigned acc = 0;
  // row blocks of this wave: block-cyclic over (workgroup, wave)
  for (size_t rb = (size_t)blockIdx.x * nw + wave; rb < blocks; rb += (size_t)gridDim.x * nw) {
    const char* base = A + (rb * RPW + (lane >> 3)) * pitch + (lane & 7u) * 16;
    auto issue = [&](int kt) {
      char* dst = ring + (kt % DEPTH) * STG;
#pragma unroll
      for (int p = 0; p < PCS; ++p)
        __builtin_amdgcn_global_load_lds((gptr_t*)(base + (size_t)p * 8 * pitch + (size_t)kt * 128), (lptr_t*)(dst + p * 1024), 16, 0, 2);
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s)
      if (s < nkt) issue(s);
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + DEPTH - 1 < nkt) {
        issue(kt + DEPTH - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PCS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const char* src = ring + (kt % DEPTH) * STG;
      u4 a = *reinterpret_cast<const u4*>(src + lane * 16);
      u4 b = *reinterpret_cast<const u4*>(src + ((lane * 16 + 512) & (STG - 1)));
      unsigned x = a[0] ^ b[1];
#pragma unroll
      for (int v = 0; v < VALU; ++v) x = x * 0x9E3779B1u + (a[v & 3] ^ b[(v + 1) & 3]);
      acc ^= x;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // C piece: RPW rows x 128 B
#pragma unroll
    for (int p = 0; p < PCS; ++p)
      __builtin_nontemporal_store(u4{acc, acc, acc, acc}, reinterpret_cast<u4*>(C + (rb * RPW + p * 8 + (lane >> 3)) * 128 + (lane & 7u) * 16));
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH, int VALU, int RPW>
static void run(const char* A, size_t pitch, size_t rows, int waves, int grid, unsigned reserve, char* C, unsigned* sink) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = reserve + (size_t)waves * DEPTH * (RPW / 8) * 1024;
  if (lds > 160 * 1024) { printf("depth %d rpw %d waves %d: LDS %zu too large\n", DEPTH, RPW, waves, lds); return; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<DEPTH, VALU, RPW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nkt = (int)(pitch / 128);
  stream<DEPTH, VALU, RPW><<<grid, waves * 64, lds>>>(A, pitch, nkt, rows, reserve, C, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  stream<DEPTH, VALU, RPW><<<grid, waves * 64, lds>>>(A, pitch, nkt, rows, reserve, C, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("pitch %5zu waves %2d grid %4d reserve %3u KiB depth %d rows/wave %2d valu %3d: %7.1f us  %6.0f GB/s of A (+ %4.0f of C)\n", pitch, waves, grid, reserve / 1024, DEPTH, RPW,
         VALU, ms * 1e3, rows * pitch / (ms * 1e-3) / 1e9, rows * 128.0 / (ms * 1e-3) / 1e9);
}

int main() {
  unsigned* sink; CK(hipMalloc(&sink, 64));
  for (size_t pitch : {(size_t)1152, (size_t)512}) {
    const size_t rows = 401408;  // 12544 x 32
    char *A, *C;
    CK(hipMalloc(&A, rows * pitch)); CK(hipMemset(A, 1, rows * pitch)); CK(hipMalloc(&C, rows * 128));
    for (int grid : {256, 512, 1568}) {
      // one 16-wave workgroup per CU, 74 KiB reserved (a resident 576 x 64 B tile)
      run<2, 64, 16>(A, pitch, rows, 16, grid, 74 * 1024, C, sink);
      run<3, 64, 16>(A, pitch, rows, 16, grid, 74 * 1024, C, sink);
      run<4, 64, 16>(A, pitch, rows, 16, grid, 74 * 1024, C, sink);
      run<2, 128, 16>(A, pitch, rows, 16, grid, 74 * 1024, C, sink);
      run<2, 64, 32>(A, pitch, rows, 16, grid, 74 * 1024, C, sink);
      run<2, 0, 16>(A, pitch, rows, 16, grid, 74 * 1024, C, sink);
    }
    // small-B case (33 KiB): two 16-wave workgroups per CU fit
    run<2, 64, 16>(A, pitch, rows, 16, 512, 40 * 1024, C, sink);
    run<3, 64, 16>(A, pitch, rows, 16, 512, 32 * 1024, C, sink);
    CK(hipFree(A)); CK(hipFree(C));
  }
  return 0;
}
