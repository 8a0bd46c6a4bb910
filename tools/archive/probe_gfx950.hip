// Hardware probe (diagnostic tool, not product code).  Determines, with exact small-integer
// data, the lane/element maps of the gfx950 sparse MFMA instructions (v_smfmac_f32_16x16x64_f16,
// v_smfmac_f32_32x32x32_f16) and of ds_read_b64_tr_b16.  The CDNA4 ISA document is not available
// offline, so the spmma kernel's operand packing is pinned by this program's output
// (a copy is committed under profiles/).
//
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_gfx950.hip -o tools/probe_gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);     \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// One wave per block.  Block `combo` encodes (L = the lane holding the single non-zero A value,
// e = which of its 8 compressed slots, p = the 2-bit position code stored in slot e's index
// field).  B: lane l element j holds the unique value l*16 + j + 1 (<= 1024, exact in fp16), so
// each non-zero output names the B (lane, element) the hardware multiplied with.
template <int SHAPE, int ABID>
__global__ void probe_smfmac(float* out) {
  const int l = threadIdx.x;
  const int combo = blockIdx.x;
  const int L = combo / 32, e = (combo / 4) % 8, p = combo % 4;
  h8 a;
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)0.0f;
  h16 b;
  for (int j = 0; j < 16; ++j) b[j] = (_Float16)(float)(l * 16 + j + 1);
  int idx = 0;
  if (l == L) {
    a[e] = (_Float16)1.0f;
    idx = (p << (2 * e));
    if (ABID) idx <<= 16;
  }
  if constexpr (SHAPE == 16) {
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, idx, 0, ABID);
    for (int r = 0; r < 4; ++r) out[(size_t)combo * 1024 + l * 4 + r] = c[r];
  } else {
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = 0;
    c = __builtin_amdgcn_smfmac_f32_32x32x32_f16(a, b, c, idx, 0, ABID);
    for (int r = 0; r < 16; ++r) out[(size_t)combo * 1024 + l * 16 + r] = c[r];
  }
}

// ds_read_b64_tr_b16: LDS word i holds the value i; lane l supplies byte address addr[l];
// dump the 4 shorts each lane receives.
__global__ void probe_tr(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) char lchar;
  typedef __attribute__((address_space(3))) s4 ls4;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ls4*)((lchar*)lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

// dense v_mfma_f32_16x16x32_f16 against the documented A/B/C maps, asymmetric integer data.
__global__ void probe_mfma16(const _Float16* A /*16x32 row-major*/,
                             const _Float16* B /*32x16 row-major*/, float* C /*16x16*/) {
  const int l = threadIdx.x;
  h8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

template <int SHAPE, int ABID>
static void run_smfmac() {
  const int ncombo = 64 * 8 * 4;
  const int nreg = SHAPE == 16 ? 4 : 16;
  const int lanes_per = SHAPE == 16 ? 16 : 32;  // lanes per k-group (= rows = cols)
  float* d;
  CK(hipMalloc(&d, (size_t)ncombo * 1024 * 4));
  CK(hipMemset(d, 0, (size_t)ncombo * 1024 * 4));
  probe_smfmac<SHAPE, ABID><<<ncombo, 64>>>(d);
  CK(hipDeviceSynchronize());
  std::vector<float> h((size_t)ncombo * 1024);
  CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  CK(hipFree(d));
  printf("== v_smfmac_f32_%s_f16 abid=%d ==\n", SHAPE == 16 ? "16x16x64" : "32x32x32", ABID);
  // Compact table: for A lane group g (row 0 of the group) and compressed slot e, which B
  // (lane group, element) is multiplied for position codes p = 0 and p = 3; then a full check
  // of hypothesis H2 over every (L, e, p).
  //   H2: A lane l: row l%R, g = l/R (R = 16 or 32); slot e, code p -> dense k = KG*g + 4*(e/2) + p
  //       with KG = 16 dense k per A lane group;
  //       B lane l: col l%R, gb = l/R; element j -> k = 8*gb + j (j<8), K/2 + 8*gb + (j-8) (j>=8).
  const int ngrp = 64 / lanes_per;          // 4 or 2 lane groups
  const int K = SHAPE == 16 ? 64 : 32;
  int bad = 0;
  for (int combo = 0; combo < ncombo; ++combo) {
    const int L = combo / 32, e = (combo / 4) % 8, p = combo % 4;
    int row_seen = -1, multi = 0, cnt = 0, bg = -1, be = -1;
    bool consistent = true;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < nreg; ++r) {
        const float v = h[(size_t)combo * 1024 + l * nreg + r];
        if (v == 0) continue;
        int row, col;
        if (SHAPE == 16) { col = l & 15; row = (l >> 4) * 4 + r; }
        else { col = l & 31; row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); }
        const int code = (int)v - 1, blane = code / 16, belem = code % 16;
        ++cnt;
        if (row_seen == -1) row_seen = row; else if (row_seen != row) multi = 1;
        if ((blane % lanes_per) != col) consistent = false;
        if (bg == -1) { bg = blane / lanes_per; be = belem; }
        else if (bg != blane / lanes_per || be != belem) consistent = false;
      }
    const int g = L / lanes_per;
    if ((L % lanes_per) == 0 && (p == 0 || p == 3))
      printf("  A(g=%d, slot %d, code %d) x B(group %d, elem %2d)   [nonzeros=%d row=%d clean=%d]\n", g, e, p, bg, be,
             cnt, row_seen, (int)(consistent && !multi));
    const int k = 16 * g + 4 * (e / 2) + p;   // dense k under H2 (A side)
    int want_bg, want_be;
    if (k < K / 2) { want_bg = k / 8; want_be = k % 8; } else { want_bg = (k - K / 2) / 8; want_be = 8 + (k % 8); }
    if (ngrp == 2 && 0) {}
    const bool ok = cnt == lanes_per && !multi && consistent && row_seen == L % lanes_per && bg == want_bg && be == want_be;
    if (!ok) ++bad;
  }
  printf("H2 mismatches (%s 16-bit half of idx): %d of %d\n", ABID ? "HIGH" : "LOW", bad, ncombo);
}

static void run_tr() {
  printf("== ds_read_b64_tr_b16 ==\n");
  // Hypothesis (guide T10): per 16-lane group, lane 4q+p supplies &blk[q][4p]; lane i receives
  // {blk[0][i], blk[1][i], blk[2][i], blk[3][i]}.  LDS image: rows of 64 shorts (128 B).
  const int row_shorts = 64;
  std::vector<int> addr(64);
  for (int l = 0; l < 64; ++l) {
    const int grp = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    // group grp reads block rows 4*grp..4*grp+3 (k) x cols 16..31
    addr[l] = 2 * ((4 * grp + q) * row_shorts + 16 + 4 * p);
  }
  int* da; short* dout;
  CK(hipMalloc(&da, 256)); CK(hipMalloc(&dout, 512));
  CK(hipMemcpy(da, addr.data(), 256, hipMemcpyHostToDevice));
  probe_tr<<<1, 64>>>(da, dout);
  CK(hipDeviceSynchronize());
  std::vector<short> out(256);
  CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int grp = l >> 4, i = l & 15;
    for (int j = 0; j < 4; ++j) {
      const int want = (4 * grp + j) * row_shorts + 16 + i;
      if (out[l * 4 + j] != want) {
        ++bad;
        if (bad <= 16) printf("  lane %d elem %d: got word %d (row %d col %d) want %d\n", l, j, out[l * 4 + j],
                              out[l * 4 + j] / row_shorts, out[l * 4 + j] % row_shorts, want);
      }
    }
  }
  printf("T10 hypothesis mismatches: %d of 256\n", bad);
  printf("raw lane0..3,16,17: ");
  for (int l : {0, 1, 2, 3, 16, 17}) printf("[%d %d %d %d] ", out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
  printf("\n");
  CK(hipFree(da)); CK(hipFree(dout));
}

static void run_mfma16() {
  printf("== v_mfma_f32_16x16x32_f16 dense map check ==\n");
  std::vector<_Float16> A(16 * 32), B(32 * 16);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (_Float16)(float)((i * 3 + k * 5) % 7 - 3);
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (_Float16)(float)((k * 2 + j * 11) % 9 - 4);
  _Float16 *dA, *dB; float* dC;
  CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, 1024));
  CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
  probe_mfma16<<<1, 64>>>(dA, dB, dC);
  CK(hipDeviceSynchronize());
  std::vector<float> C(256);
  CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float s = 0;
    for (int k = 0; k < 32; ++k) s += (float)A[i * 32 + k] * (float)B[k * 16 + j];
    if (s != C[i * 16 + j]) ++bad;
  }
  printf("dense 16x16x32 f16 mismatches: %d of 256\n", bad);
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch %s CUs %d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  run_mfma16();
  run_smfmac<16, 0>();
  run_smfmac<16, 1>();
  run_smfmac<32, 0>();
  run_smfmac<32, 1>();
  run_tr();
  return 0;
}
