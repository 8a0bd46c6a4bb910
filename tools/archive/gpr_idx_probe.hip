// gpr_idx_probe.hip -- can a B chunk live in registers and be indexed per non-zero?  
// Every lane holds 64 "columns" x 2 floats in v[130:257)-style fixed registers (here v[128:255]); entries {index word, value}
// sit in LDS and reach all lanes by broadcast reads; per entry: v_readfirstlane m0 <- index word (column * 2 | SRC0_REL mode
// bit), NOPS wait states, v_pk_fma_f32 acc, v[128:129](+M0), value (VGPR pair, high half for both lanes), acc.
// Prints correctness against a host sum and ns per entry with two waves per SIMD.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CL128 "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255"
#define FMA(REG) "v_pk_fma_f32 v[102:103], v[" #REG ":" #REG "+1], v[128:129], v[102:103] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
#define FMB(REG) "v_pk_fma_f32 v[98:99], v[" #REG ":" #REG "+1], v[128:129], v[98:99] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n"
// variant 2: two accumulator chains; variant 3: two chains, the four indices fetched up front (v_readfirstlane cannot write m0 itself)
#define GROUP2(R0, R1, R2, R3) \
  RFL(s21, R0) RFL(s22, R1) SETM(s21, 0) FMA(R0) RFL(s21, R2) SETM(s22, 0) FMB(R1) RFL(s22, R3) SETM(s21, 0) FMA(R2) SETM(s22, 0) FMB(R3)
#define GROUP3(R0, R1, R2, R3) \
  RFL(s21, R0) RFL(s22, R1) RFL(s23, R2) RFL(s24, R3) SETM(s21, 0) FMA(R0) SETM(s22, 0) FMB(R1) SETM(s23, 0) FMA(R2) SETM(s24, 0) FMB(R3)
#define RFL(S, REG) "v_readfirstlane_b32 " #S ", v" #REG "\n"
#define SETM(S, NOPS) "s_mov_b32 m0, " #S "\n s_nop " #NOPS "\n"
// four entries in registers R0..R0+7: the next index is fetched while the previous FMA issues (M0 itself is single)
#define GROUP(R0, R1, R2, R3, NOPS) \
  RFL(s21, R0) RFL(s22, R1) SETM(s21, NOPS) FMA(R0) RFL(s21, R2) SETM(s22, NOPS) FMA(R1) RFL(s22, R3) SETM(s21, NOPS) FMA(R2) SETM(s22, NOPS) FMA(R3)
template <int NOPS>
__global__ __launch_bounds__(64) void probe(int ngroups, const unsigned* ent, float* out) {
  extern __shared__ unsigned lds[];
  const unsigned lane = threadIdx.x;
  for (int i = lane; i < ngroups * 8; i += 64) lds[i] = ent[i];
  __syncthreads();
  float base = (float)lane * 0.001f, a0, a1;
  asm volatile(
      "s_mov_b32 s20, 0\n"
      "1:\n"
      "v_cvt_f32_u32 v100, s20\n"
      "v_add_f32 v100, v100, %[base]\n"
      "s_set_gpr_idx_on s20, 8\n"
      "v_mov_b32 v128, v100\n"
      "s_set_gpr_idx_off\n"
      "s_add_u32 s20, s20, 1\n"
      "s_cmp_lt_u32 s20, 128\n"
      "s_cbranch_scc1 1b\n"
      :: [base] "v"(base) : "s20", "v100", CL128, "scc", "memory");
  // groups of 4 entries = 32 bytes; set A = v[104:111], set B = v[112:119]; two-stage software pipeline
  if (NOPS == 0)
  asm volatile(
      "v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v101, 0\n"
      "s_mov_b32 s20, %[n]\n s_mov_b32 s21, 0\n s_set_gpr_idx_on s21, 2\n"
      "ds_read_b128 v[104:107], v101\n ds_read_b128 v[108:111], v101 offset:16\n"
      "2:\n"
      "ds_read_b128 v[112:115], v101 offset:32\n ds_read_b128 v[116:119], v101 offset:48\n"
      "s_waitcnt lgkmcnt(2)\n"
      GROUP(104, 106, 108, 110, 0)
      "s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 3f\n"
      "ds_read_b128 v[104:107], v101 offset:64\n ds_read_b128 v[108:111], v101 offset:80\n"
      "v_add_u32_e64 v101, v101, 64\n"
      "s_waitcnt lgkmcnt(2)\n"
      GROUP(112, 114, 116, 118, 0)
      "s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc0 2b\n"
      "3:\n s_waitcnt lgkmcnt(0)\n s_set_gpr_idx_off\n s_mov_b32 m0, -1\n"
      "v_mov_b32 %[a0], v102\n v_mov_b32 %[a1], v103\n"
      : [a0] "=v"(a0), [a1] "=v"(a1) : [n] "s"(ngroups)
      : "s20", "s21", "s22", "s23", "s24", "m0", "v101", "v102", "v103", "v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", "scc", "memory");
  else if (NOPS == 2)
  asm volatile(
      "v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v98, 0\n v_mov_b32 v99, 0\n v_mov_b32 v101, 0\n"
      "s_mov_b32 s20, %[n]\n s_mov_b32 s21, 0\n s_set_gpr_idx_on s21, 2\n"
      "ds_read_b128 v[104:107], v101\n ds_read_b128 v[108:111], v101 offset:16\n"
      "2:\n"
      "ds_read_b128 v[112:115], v101 offset:32\n ds_read_b128 v[116:119], v101 offset:48\n"
      "s_waitcnt lgkmcnt(2)\n"
      GROUP2(104, 106, 108, 110)
      "s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 3f\n"
      "ds_read_b128 v[104:107], v101 offset:64\n ds_read_b128 v[108:111], v101 offset:80\n"
      "v_add_u32_e64 v101, v101, 64\n"
      "s_waitcnt lgkmcnt(2)\n"
      GROUP2(112, 114, 116, 118)
      "s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc0 2b\n"
      "3:\n s_waitcnt lgkmcnt(0)\n s_set_gpr_idx_off\n s_mov_b32 m0, -1\n"
      "v_add_f32 %[a0], v102, v98\n v_add_f32 %[a1], v103, v99\n"
      : [a0] "=v"(a0), [a1] "=v"(a1) : [n] "s"(ngroups)
      : "s20", "s21", "s22", "s23", "s24", "m0", "v98", "v99", "v101", "v102", "v103", "v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", "scc", "memory");
  else
  asm volatile(
      "v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v98, 0\n v_mov_b32 v99, 0\n v_mov_b32 v101, 0\n"
      "s_mov_b32 s20, %[n]\n s_mov_b32 s21, 0\n s_set_gpr_idx_on s21, 2\n"
      "ds_read_b128 v[104:107], v101\n ds_read_b128 v[108:111], v101 offset:16\n"
      "2:\n"
      "ds_read_b128 v[112:115], v101 offset:32\n ds_read_b128 v[116:119], v101 offset:48\n"
      "s_waitcnt lgkmcnt(2)\n"
      GROUP3(104, 106, 108, 110)
      "s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc1 3f\n"
      "ds_read_b128 v[104:107], v101 offset:64\n ds_read_b128 v[108:111], v101 offset:80\n"
      "v_add_u32_e64 v101, v101, 64\n"
      "s_waitcnt lgkmcnt(2)\n"
      GROUP3(112, 114, 116, 118)
      "s_sub_u32 s20, s20, 1\n s_cmp_eq_u32 s20, 0\n s_cbranch_scc0 2b\n"
      "3:\n s_waitcnt lgkmcnt(0)\n s_set_gpr_idx_off\n s_mov_b32 m0, -1\n"
      "v_add_f32 %[a0], v102, v98\n v_add_f32 %[a1], v103, v99\n"
      : [a0] "=v"(a0), [a1] "=v"(a1) : [n] "s"(ngroups)
      : "s20", "s21", "s22", "s23", "s24", "m0", "v98", "v99", "v101", "v102", "v103", "v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119", "scc", "memory");
  out[(size_t)blockIdx.x * 128 + 2 * lane] = a0;
  out[(size_t)blockIdx.x * 128 + 2 * lane + 1] = a1;
}
template <int NOPS>
static void run(int ngroups, const unsigned* d_ent, float* d_out, const std::vector<unsigned>& h_ent) {
  const size_t lds = (size_t)ngroups * 32 + 64;
  probe<NOPS><<<1, 64, lds>>>(ngroups, d_ent, d_out);
  (void)hipDeviceSynchronize();
  float h[128];
  (void)hipMemcpy(h, d_out, 512, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0;
  for (int i = 0; i < ngroups * 4; ++i) {
    const unsigned idx = h_ent[2 * i] & 0xff;
    float v; memcpy(&v, &h_ent[2 * i + 1], 4);
    e0 += (double)v * idx; e1 += (double)v * (idx + 1);
  }
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  printf("variant %d (0: one chain, 2: two accumulator chains, 3: two chains, the group's four indices fetched up front): lane 0 got %.1f %.1f expect %.1f %.1f\n", NOPS, h[0], h[1], e0, e1);
  for (int waves : {1024, 2048}) {   // one / two waves per SIMD (256 VGPRs each)
    probe<NOPS><<<waves, 64, lds>>>(ngroups, d_ent, d_out);
    (void)hipEventRecord(a);
    for (int r = 0; r < 20; ++r) probe<NOPS><<<waves, 64, lds>>>(ngroups, d_ent, d_out);
    (void)hipEventRecord(b); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double per = ms / 20 * 1e6 / (ngroups * 4.0);
    printf("   %d waves (%d per SIMD): %.2f ns per entry per wave = %.2f ns per entry per SIMD\n", waves, waves / 1024, per, per / (waves / 1024));
  }
}
int main() {
  const int ngroups = 512;  // 2048 entries, 16 KiB of LDS: eight single-wave workgroups per CU, two per SIMD
  std::vector<unsigned> ent(ngroups * 8);
  unsigned idx = 1;
  for (int i = 0; i < ngroups * 4; ++i) {
    idx = (5 * idx + 3) & 63;
    ent[2 * i] = (2 * idx) | 0x2000u;  // M0: index | SRC1_REL mode bit
    float v = (float)((i % 5) + 1);
    memcpy(&ent[2 * i + 1], &v, 4);
  }
  unsigned* d_ent; float* d_out;
  (void)hipMalloc(&d_ent, ent.size() * 4); (void)hipMalloc(&d_out, 2048 * 512);
  (void)hipMemcpy(d_ent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice);
  run<0>(ngroups, d_ent, d_out, ent);
  run<2>(ngroups, d_ent, d_out, ent);
  run<3>(ngroups, d_ent, d_out, ent);
  return 0;
}
