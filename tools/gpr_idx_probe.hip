#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// v_pk_fma_f32 with a relative-indexed src0 (VGPR indexing mode): correctness and cycles per entry
__global__ __launch_bounds__(64) void probe(int n, float* out, unsigned long long* cyc, const unsigned* idxs) {
  const unsigned lane = threadIdx.x;
  float base = (float)lane * 0.001f;
  float a0 = 0.f, a1 = 0.f;
  unsigned long long t0, t1;
  // fill v[128 + i] = i + base for i in 0..127 through DST-relative moves
  asm volatile(
      "s_mov_b32 s20, 0\n"
      "1:\n"
      "v_cvt_f32_u32 v100, s20\n"
      "v_add_f32 v100, v100, %[base]\n"
      "s_set_gpr_idx_on s20, 8\n"      // DST_REL
      "v_mov_b32 v128, v100\n"
      "s_set_gpr_idx_off\n"
      "s_add_u32 s20, s20, 1\n"
      "s_cmp_lt_u32 s20, 128\n"
      "s_cbranch_scc1 1b\n"
      :: [base] "v"(base) : "s20", "v100", "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255", "scc", "memory");
  // entries: idx = (5 * idx + 3) & 63 (pair index), value 1.0
  asm volatile(
      "s_memtime %[t0]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_mov_b32 s20, 0\n"       // counter
      "s_mov_b32 s21, 1\n"       // idx
      "s_mov_b32 s22, 1.0\n"
      "s_mov_b32 s23, 1.0\n"
      "v_mov_b32 v102, 0\n"
      "v_mov_b32 v103, 0\n"
      "s_set_gpr_idx_on s21, 1\n"   // SRC0_REL
      "2:\n"
      "s_mul_i32 s21, s21, 5\n"
      "s_add_u32 s21, s21, 3\n"
      "s_and_b32 s21, s21, 63\n"
      "s_lshl_b32 s24, s21, 1\n"
      "s_set_gpr_idx_idx s24\n"
      "v_pk_fma_f32 v[102:103], v[128:129], s[22:23], v[102:103]\n"
      "s_add_u32 s20, s20, 1\n"
      "s_cmp_lt_u32 s20, %[n]\n"
      "s_cbranch_scc1 2b\n"
      "s_set_gpr_idx_off\n"
      "s_memtime %[t1]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_mov_b32 %[a0], v102\n"
      "v_mov_b32 %[a1], v103\n"
      : [t0] "=&s"(t0), [t1] "=&s"(t1), [a0] "=v"(a0), [a1] "=v"(a1)
      : [n] "s"(n)
      : "s20", "s21", "s22", "s23", "s24", "v102", "v103", "scc", "memory");
  out[2 * lane] = a0;
  out[2 * lane + 1] = a1;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  (void)idxs;
}
int main() {
  const int n = 4096;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 128 * 4); hipMalloc(&cyc, 8 * 1024);
  hipMemset(out, 0, 512);
  // v102/v103 must start at zero: (they are clobbers; initial content undefined) -> run twice and difference? simply report
  probe<<<1, 64>>>(n, out, cyc, nullptr);
  hipDeviceSynchronize();
  float h[128]; unsigned long long c;
  hipMemcpy(h, out, 512, hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // expected: sum over entries of v[128 + 2*idx] (+0.001*lane per entry) for pair lo, v[129 + 2*idx] for hi
  double e0 = 0, e1 = 0; unsigned idx = 1;
  for (int i = 0; i < n; ++i) { idx = (5 * idx + 3) & 63; e0 += 2 * idx; e1 += 2 * idx + 1; }
  printf("lane0: got %.1f %.1f expect(+init) %.1f %.1f ; lane 5: %.2f (expect %.2f)\n", h[0], h[1], e0, e1, h[10], e0 + 5 * 0.001 * n);
  printf("memtime ticks %llu for %d entries (100 MHz ticks?)\n", c, n);
  // many waves for throughput: 256 CUs x 8 waves
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<<<2048, 64>>>(1 << 16, out, cyc, nullptr);
  hipEventRecord(a); probe<<<2048, 64>>>(1 << 16, out, cyc, nullptr); hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("2048 waves x 65536 entries: %.3f ms -> %.2f ns per entry per wave; at 2 waves/SIMD: %.2f ns per entry per SIMD\n", ms, ms * 1e6 / 65536, ms * 1e6 / 65536 / 2);
  return 0;
}
