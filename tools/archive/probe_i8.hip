// Hardware probe (diagnostic tool, not product code): operand maps of v_smfmac_i32_16x16x128_i8 on gfx950, found with
// exact integer data.  One wave per block; block = (L, e, p): lane L holds a single non-zero A byte (value 1) in
// compressed slot e (0..15) whose 2-bit index code is p.  B lane l byte j holds l (pass 0) or j (pass 1), so every
// non-zero output names the B (lane, byte) the hardware multiplied with.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_i8.hip -o tools/probe_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void probe(int* out, int pass) {
  const int l = threadIdx.x, combo = blockIdx.x;
  const int L = combo / 64, e = (combo / 4) % 16, p = combo % 4;
  unsigned char ab[16] = {0};
  unsigned char bb[32];
  for (int j = 0; j < 32; ++j) bb[j] = (unsigned char)(pass == 0 ? l : j);
  int idx = 0;
  if (l == L) { ab[e] = 1; idx = p << (2 * e); }
  i4 a; i8v b;
  __builtin_memcpy(&a, ab, 16);
  __builtin_memcpy(&b, bb, 32);
  i4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a, b, c, idx, 0, 0);
  for (int q = 0; q < 4; ++q) out[((size_t)combo * 64 + l) * 4 + q] = c[q];
}

int main() {
  const int combos = 64 * 16 * 4;
  int* d; CK(hipMalloc(&d, (size_t)combos * 64 * 4 * sizeof(int)));
  std::vector<int> h0((size_t)combos * 256), h1((size_t)combos * 256);
  probe<<<combos, 64>>>(d, 0); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h0.data(), d, h0.size() * 4, hipMemcpyDeviceToHost));
  probe<<<combos, 64>>>(d, 1); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h1.data(), d, h1.size() * 4, hipMemcpyDeviceToHost));
  printf("v_smfmac_i32_16x16x128_i8: A lane L (row?, group g = L / 16), slot e, code p  ->  output lanes / B operand\n");
  for (int combo = 0; combo < combos; ++combo) {
    const int L = combo / 64, e = (combo / 4) % 16, p = combo % 4;
    if ((L & 15) != 3 && (L & 15) != 0) continue;       // rows 0 and 3 of every group are enough
    if (p != 0 && p != 3 && !(e < 2)) continue;
    int nz = 0, first_lane = -1, first_q = -1, bl = -1, bj = -1, consistent = 1;
    for (int l = 0; l < 64; ++l)
      for (int q = 0; q < 4; ++q) {
        const int v0 = h0[((size_t)combo * 64 + l) * 4 + q], v1 = h1[((size_t)combo * 64 + l) * 4 + q];
        if (v0 != 0 || v1 != 0) {
          ++nz;
          const int col = l & 15;                         // output column if the D map is the f16 one
          if (first_lane < 0) { first_lane = l; first_q = q; bl = v0; bj = v1; }
          if (v1 != bj || ((v0 - col) & 15) != ((bl - (first_lane & 15)) & 15)) consistent = 0;
        }
      }
    printf("A(L=%2d g=%d row=%2d, slot %2d, code %d): nonzeros %2d  first at lane %2d reg %d (-> out row %2d?)  B lane %2d (group %d) byte %2d  consistent=%d\n",
           L, L / 16, L & 15, e, p, nz, first_lane, first_q, first_lane < 0 ? -1 : 4 * (first_lane >> 4) + first_q, bl, bl >> 4, bj, consistent);
  }
  return 0;
}
