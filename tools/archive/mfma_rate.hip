// Microbenchmark (diagnostic tool): issue rate of the matrix instructions the kernels use on gfx950,
//   v_smfmac_f32_16x16x64_f16 (2:4 sparse), v_mfma_f32_16x16x32_f16 (dense), v_mfma_f32_16x16x4_f32,
// as cycles per instruction per SIMD with 1 / 2 waves per SIMD and 4 / 8 independent accumulators per wave.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int KIND, int NACC>
__global__ void rate(int iters, float* sink, unsigned long long* cyc) {
  f4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  h8 a8; h16 b16;
#pragma unroll
  for (int i = 0; i < 8; ++i) a8[i] = (_Float16)(threadIdx.x * 0.001f + i);
#pragma unroll
  for (int i = 0; i < 16; ++i) b16[i] = (_Float16)(threadIdx.x * 0.002f - i);
  const int idx = 0x4444 * (threadIdx.x & 1) + 0x4e4e;
  const float fa = threadIdx.x * 0.5f, fb = 1.f - threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a8, b16, acc[i], idx, 0, 0);
      if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, acc[i], 0, 0, 0);
      if constexpr (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NACC>
static void run(const char* name, int waves_per_cu, double flops_per_instr) {
  float* sink; unsigned long long* cyc;
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, 64));
  const int iters = 20000, blocks = 256 * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate<KIND, NACC><<<blocks, waves_per_cu * 64>>>(100, sink, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate<KIND, NACC><<<blocks, waves_per_cu * 64>>>(iters, sink, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double instr = (double)blocks * waves_per_cu * iters * NACC;
  printf("%-28s waves/WG %2d acc %d : %7.2f ms  %8.1f TF/s (as named flops)  wave-0 clk/instr %.2f (s_memtime/realtime ticks, see note)\n",
         name, waves_per_cu, NACC, ms, instr * flops_per_instr / (ms * 1e-3) / 1e12, (double)c / ((double)iters * NACC));
  CK(hipFree(sink)); CK(hipFree(cyc));
}

int main() {
  printf("note: TF/s is the chip-wide rate (1024 workgroups, 4 per CU when they fit); flops counted dense-equivalent\n");
  run<0, 4>("smfmac_f32_16x16x64_f16", 4, 2.0 * 16 * 16 * 64);
  run<0, 8>("smfmac_f32_16x16x64_f16", 4, 2.0 * 16 * 16 * 64);
  run<0, 8>("smfmac_f32_16x16x64_f16", 8, 2.0 * 16 * 16 * 64);
  run<1, 4>("mfma_f32_16x16x32_f16", 4, 2.0 * 16 * 16 * 32);
  run<1, 8>("mfma_f32_16x16x32_f16", 4, 2.0 * 16 * 16 * 32);
  run<1, 8>("mfma_f32_16x16x32_f16", 8, 2.0 * 16 * 16 * 32);
  run<2, 8>("mfma_f32_16x16x4_f32", 4, 2.0 * 16 * 16 * 4);
  run<2, 8>("mfma_f32_16x16x4_f32", 8, 2.0 * 16 * 16 * 4);
  return 0;
}
