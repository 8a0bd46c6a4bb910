// tools/hipsparselt_yardstick.cpp -- YARDSTICK ONLY (never linked by sparsify.me_amd/, tests' pass/fail or bench.py).
//
// The reference delegates prune / check / compress / matmul to closed cuSPARSELt (include/sparsify.me/spmma.hxx:86-113)
// and holds no fixtures, so the tie-breaks and the TILE rule of this build are frozen by its own oracle.  The one
// independent implementation of the same vendor contract on the box is /opt/rocm/lib/libhipsparselt.so: this tool puts
// it next to the sm_* entry points and reports
//   (a) semantics: the fraction of 1x4 strips (STRIP) / 4x4 tiles (TILE) in which hipsparseLtSpMMAPrune keeps exactly
//       the elements sm_prune24_f16 keeps, on seeded U(0,1) data and on tie-heavy small-integer data, plus the
//       disagreement classes (same L1 norm kept / different norm), and PruneCheck agreement;
//   (b) time: hipsparseLtMatmul on its own compressed operand against sm_spmma_f16 on the blob, per shape, b = 32,
//       both timed the same way (HIP events around REPS launches cycling NSET operand sets, after warm-up).
// It cannot raise the parity grade (it is not the reference); it is third-party evidence for the frozen rules.
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/hipsparselt_yardstick.cpp -Iinclude -I/opt/rocm/include \
//          -L/opt/rocm/lib -lhipsparselt -Lsparsify.me_amd -lsparsifyme -Wl,-rpath,/opt/rocm/lib -o tools/hipsparselt_yardstick
// run:   LD_LIBRARY_PATH=sparsify.me_amd tools/hipsparselt_yardstick [table.csv] [batch]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipsparselt/hipsparselt.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "sparsifyme.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define CKS(x) do { hipsparseStatus_t s_ = (x); if (s_ != HIPSPARSE_STATUS_SUCCESS) { fprintf(stderr, "hipsparselt status %d at line %d (%s)\n", (int)s_, __LINE__, #x); return false; } } while (0)
#define CKM(x) do { int s_ = (x); if (s_ != 0) { fprintf(stderr, "sm status %d at line %d: %s\n", s_, __LINE__, sm_last_error()); exit(2); } } while (0)

static float h2f(uint16_t h) { __half x; memcpy(&x, &h, 2); return __half2float(x); }
static uint16_t f2h(float f) { __half x = __float2half(f); uint16_t h; memcpy(&h, &x, 2); return h; }

struct Vendor {
  hipsparseLtHandle_t handle;
  hipsparseLtMatDescriptor_t matA, matB, matC;
  hipsparseLtMatmulDescriptor_t mm;
  hipsparseLtMatmulAlgSelection_t alg;
  hipsparseLtMatmulPlan_t plan;
  bool have_plan = false;
  // A: M x K structured, B: K x N dense, C: M x N, all row-major (the reference's layout, spmma.hxx:40-64)
  bool init(int64_t M, int64_t N, int64_t K) {
    CKS(hipsparseLtInit(&handle));
    CKS(hipsparseLtStructuredDescriptorInit(&handle, &matA, M, K, K, 16, HIP_R_16F, HIPSPARSE_ORDER_ROW, HIPSPARSELT_SPARSITY_50_PERCENT));
    CKS(hipsparseLtDenseDescriptorInit(&handle, &matB, K, N, N, 16, HIP_R_16F, HIPSPARSE_ORDER_ROW));
    CKS(hipsparseLtDenseDescriptorInit(&handle, &matC, M, N, N, 16, HIP_R_16F, HIPSPARSE_ORDER_ROW));
    CKS(hipsparseLtMatmulDescriptorInit(&handle, &mm, HIPSPARSE_OPERATION_NON_TRANSPOSE, HIPSPARSE_OPERATION_NON_TRANSPOSE, &matA, &matB, &matC, &matC, HIPSPARSELT_COMPUTE_32F));
    CKS(hipsparseLtMatmulAlgSelectionInit(&handle, &alg, &mm, HIPSPARSELT_MATMUL_ALG_DEFAULT));
    CKS(hipsparseLtMatmulPlanInit(&handle, &plan, &mm, &alg));
    have_plan = true;
    return true;
  }
  void destroy() {
    if (have_plan) hipsparseLtMatmulPlanDestroy(&plan);
    hipsparseLtMatDescriptorDestroy(&matA);
    hipsparseLtMatDescriptorDestroy(&matB);
    hipsparseLtMatDescriptorDestroy(&matC);
    hipsparseLtDestroy(&handle);
  }
};

struct Agree {
  size_t units = 0, same = 0, same_norm = 0, diff_norm = 0, vendor_invalid = 0;
};

// per 1x4 strip: do the two pruned outputs coincide?  if not: same kept L1 norm (a tie broken differently) or not
static Agree compare_strips(const std::vector<uint16_t>& a, const std::vector<uint16_t>& v, size_t M, size_t K) {
  Agree g;
  for (size_t r = 0; r < M; ++r)
    for (size_t c = 0; c + 4 <= K; c += 4) {
      const uint16_t* x = &a[r * K + c];
      const uint16_t* y = &v[r * K + c];
      ++g.units;
      if (!memcmp(x, y, 8)) { ++g.same; continue; }
      float nx = 0, ny = 0; int nzy = 0;
      for (int i = 0; i < 4; ++i) { nx += fabsf(h2f(x[i])); ny += fabsf(h2f(y[i])); nzy += (y[i] & 0x7fff) != 0; }
      if (nzy > 2) ++g.vendor_invalid;
      if (nx == ny) ++g.same_norm; else ++g.diff_norm;
    }
  return g;
}
static Agree compare_tiles(const std::vector<uint16_t>& a, const std::vector<uint16_t>& v, size_t M, size_t K) {
  Agree g;
  for (size_t r = 0; r + 4 <= M; r += 4)
    for (size_t c = 0; c + 4 <= K; c += 4) {
      ++g.units;
      bool same = true; double nx = 0, ny = 0; bool bad = false;
      int colcnt[4] = {0, 0, 0, 0};
      for (int i = 0; i < 4; ++i) {
        const uint16_t* x = &a[(r + i) * K + c];
        const uint16_t* y = &v[(r + i) * K + c];
        if (memcmp(x, y, 8)) same = false;
        int rowcnt = 0;
        for (int j = 0; j < 4; ++j) { nx += fabs((double)h2f(x[j])); ny += fabs((double)h2f(y[j])); if (y[j] & 0x7fff) { ++rowcnt; ++colcnt[j]; } }
        if (rowcnt > 2) bad = true;
      }
      for (int j = 0; j < 4; ++j) if (colcnt[j] > 2) bad = true;
      if (same) { ++g.same; continue; }
      if (bad) ++g.vendor_invalid;
      if (nx == ny) ++g.same_norm; else ++g.diff_norm;
    }
  return g;
}

static void fill_ties(std::vector<uint16_t>& h, uint64_t seed) {  // small integers -3..3: most strips hold equal magnitudes
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
  for (auto& x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = f2h((float)((int)((s >> 33) % 7) - 3)); }
}

static bool semantics(FILE* out, size_t M, size_t K) {
  Vendor v;
  if (!v.init((int64_t)M, 64, (int64_t)K)) return false;
  uint16_t *dA, *dO, *dV; int* dflag;
  CK(hipMalloc(&dA, M * K * 2)); CK(hipMalloc(&dO, M * K * 2)); CK(hipMalloc(&dV, M * K * 2)); CK(hipMalloc(&dflag, 4));
  std::vector<uint16_t> ho(M * K), hv(M * K), hin(M * K);
  for (int data = 0; data < 2; ++data) {
    if (data == 0) { CKM(sm_fill_uniform_f16(dA, M * K, 0x5eed, 0.f, 1.f, nullptr)); }
    else { fill_ties(hin, 7); CK(hipMemcpy(dA, hin.data(), M * K * 2, hipMemcpyHostToDevice)); }
    for (int alg = 0; alg < 2; ++alg) {  // 0 = TILE, 1 = STRIP in both libraries
      CKM(sm_prune24_f16(dA, dO, M, K, K, alg, nullptr));
      CKS(hipsparseLtSpMMAPrune(&v.handle, &v.mm, dA, dV, alg == 0 ? HIPSPARSELT_PRUNE_SPMMA_TILE : HIPSPARSELT_PRUNE_SPMMA_STRIP, nullptr));
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(ho.data(), dO, M * K * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hv.data(), dV, M * K * 2, hipMemcpyDeviceToHost));
      const Agree g = alg == 1 ? compare_strips(ho, hv, M, K) : compare_tiles(ho, hv, M, K);
      // cross-check: each library's check on the other's output, and on the dense input
      int f_sm_on_v = -1, f_v_on_sm = -1, f_v_dense = -1, f_sm_dense = -1;
      CKM(sm_prune24_check_f16(dV, M, K, K, dflag, nullptr)); CK(hipMemcpy(&f_sm_on_v, dflag, 4, hipMemcpyDeviceToHost));
      CKS(hipsparseLtSpMMAPruneCheck(&v.handle, &v.mm, dO, dflag, nullptr)); CK(hipDeviceSynchronize()); CK(hipMemcpy(&f_v_on_sm, dflag, 4, hipMemcpyDeviceToHost));
      CKS(hipsparseLtSpMMAPruneCheck(&v.handle, &v.mm, dA, dflag, nullptr)); CK(hipDeviceSynchronize()); CK(hipMemcpy(&f_v_dense, dflag, 4, hipMemcpyDeviceToHost));
      CKM(sm_prune24_check_f16(dA, M, K, K, dflag, nullptr)); CK(hipMemcpy(&f_sm_dense, dflag, 4, hipMemcpyDeviceToHost));
      fprintf(out, "semantics %-5s %-7s %zux%zu: %zu units, identical %.6f, differ with the same kept L1 norm %.6f, differ with another norm %.6f, vendor output not 2:4 in %zu | check: sm(vendor out)=%d vendor(sm out)=%d vendor(dense in)=%d sm(dense in)=%d\n",
              alg == 0 ? "TILE" : "STRIP", data == 0 ? "uniform" : "ties", M, K, g.units, (double)g.same / g.units, (double)g.same_norm / g.units,
              (double)g.diff_norm / g.units, g.vendor_invalid, f_sm_on_v, f_v_on_sm, f_v_dense, f_sm_dense);
      // first few disagreeing units, for DESIGN.md's class descriptions
      int shown = 0;
      if (alg == 1) {
        for (size_t r = 0; r < M && shown < 3; ++r)
          for (size_t c = 0; c + 4 <= K && shown < 3; c += 4)
            if (memcmp(&ho[r * K + c], &hv[r * K + c], 8)) {
              if (data == 0) CK(hipMemcpy(hin.data() + r * K + c, dA + r * K + c, 8, hipMemcpyDeviceToHost));
              fprintf(out, "   strip (%zu,%zu) in [%g %g %g %g] sm [%g %g %g %g] vendor [%g %g %g %g]\n", r, c, h2f(hin[r * K + c]), h2f(hin[r * K + c + 1]), h2f(hin[r * K + c + 2]), h2f(hin[r * K + c + 3]),
                      h2f(ho[r * K + c]), h2f(ho[r * K + c + 1]), h2f(ho[r * K + c + 2]), h2f(ho[r * K + c + 3]), h2f(hv[r * K + c]), h2f(hv[r * K + c + 1]), h2f(hv[r * K + c + 2]), h2f(hv[r * K + c + 3]));
              ++shown;
            }
      } else {
        for (size_t r = 0; r + 4 <= M && shown < 2; r += 4)
          for (size_t c = 0; c + 4 <= K && shown < 2; c += 4) {
            bool same = true;
            for (int i = 0; i < 4; ++i) if (memcmp(&ho[(r + i) * K + c], &hv[(r + i) * K + c], 8)) same = false;
            if (same) continue;
            if (data == 0) for (int i = 0; i < 4; ++i) CK(hipMemcpy(hin.data() + (r + i) * K + c, dA + (r + i) * K + c, 8, hipMemcpyDeviceToHost));
            fprintf(out, "   tile (%zu,%zu):\n", r, c);
            for (int i = 0; i < 4; ++i) {
              const size_t o = (r + i) * K + c;
              fprintf(out, "      in [%7.4f %7.4f %7.4f %7.4f]  sm [%7.4f %7.4f %7.4f %7.4f]  vendor [%7.4f %7.4f %7.4f %7.4f]\n", h2f(hin[o]), h2f(hin[o + 1]), h2f(hin[o + 2]), h2f(hin[o + 3]),
                      h2f(ho[o]), h2f(ho[o + 1]), h2f(ho[o + 2]), h2f(ho[o + 3]), h2f(hv[o]), h2f(hv[o + 1]), h2f(hv[o + 2]), h2f(hv[o + 3]));
            }
            ++shown;
          }
      }
      fflush(out);
    }
  }
  CK(hipFree(dA)); CK(hipFree(dO)); CK(hipFree(dV)); CK(hipFree(dflag));
  v.destroy();
  return true;
}

struct Shape { size_t m, n, k, b; int count; };

static bool time_shape(FILE* out, const Shape& s, double& t_sm_tot, double& t_v_tot, double& t_vs_tot) {
  const size_t M = s.m * s.b, N = s.n, K = s.k;
  constexpr int NSET = 4, WARM = 3, REPS = 12;
  Vendor v;
  if (!v.init((int64_t)M, (int64_t)N, (int64_t)K)) return false;
  size_t csize = 0, cbuf = 0, blob_bytes = 0, ws = 0;
  CKS(hipsparseLtSpMMACompressedSize(&v.handle, &v.plan, &csize, &cbuf));
  CKM(sm_compress24_size(s.m, K, 2, s.b, &blob_bytes));
  uint16_t *dA[NSET], *dB, *dC[NSET], *dCv[NSET]; void *vblob[NSET], *blob[NSET], *dcbuf = nullptr, *dws = nullptr;
  CK(hipMalloc(&dB, K * N * 2));
  CKM(sm_fill_uniform_f16(dB, K * N, 99, 0.f, 1.f, nullptr));
  if (cbuf) CK(hipMalloc(&dcbuf, cbuf));
  for (int i = 0; i < NSET; ++i) {
    CK(hipMalloc(&dA[i], M * K * 2)); CK(hipMalloc(&dC[i], M * N * 2)); CK(hipMalloc(&dCv[i], M * N * 2));
    CK(hipMalloc(&vblob[i], csize)); CK(hipMalloc(&blob[i], blob_bytes));
    CKM(sm_fill_uniform_f16(dA[i], M * K, 1000 + i, 0.f, 1.f, nullptr));
    CKM(sm_prune24_f16(dA[i], dA[i], M, K, K, 1 /*STRIP*/, nullptr));  // one pruned operand for both libraries
    CKM(sm_compress24_f16(dA[i], s.m, K, K, s.b, s.m * K, blob[i], nullptr));
    CKS(hipsparseLtSpMMACompress(&v.handle, &v.plan, dA[i], vblob[i], dcbuf, nullptr));
  }
  CK(hipDeviceSynchronize());
  CKS(hipsparseLtMatmulGetWorkspace(&v.handle, &v.plan, &ws));
  if (ws) CK(hipMalloc(&dws, ws));
  const float alpha = 1.f, beta = 0.f;
  hipStream_t st = nullptr;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](auto&& fn) {
    for (int i = 0; i < WARM; ++i) fn(i % NSET);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < REPS; ++i) fn(i % NSET);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)ms * 1e3 / REPS;  // us per launch
  };
  bool vendor_ok = true;
  const double t_sm = time_it([&](int i) { CKM(sm_spmma_f16(blob[i], dB, dC[i], s.m, N, K, s.b, 0, s.m * N, 1.f, 0.f, st)); });
  const double t_v = time_it([&](int i) {
    if (hipsparseLtMatmul(&v.handle, &v.plan, &alpha, vblob[i], dB, &beta, dCv[i], dCv[i], dws, &st, 1) != HIPSPARSE_STATUS_SUCCESS) vendor_ok = false;
  });
  // the vendor's own search over its kernels for this problem, then the same timing on the plan it settles on
  double t_vs = -1;
  if (vendor_ok && hipsparseLtMatmulSearch(&v.handle, &v.plan, &alpha, vblob[0], dB, &beta, dCv[0], dCv[0], dws, &st, 1) == HIPSPARSE_STATUS_SUCCESS) {
    size_t ws2 = 0;
    if (hipsparseLtMatmulGetWorkspace(&v.handle, &v.plan, &ws2) == HIPSPARSE_STATUS_SUCCESS && ws2 > ws) { if (dws) CK(hipFree(dws)); CK(hipMalloc(&dws, ws2)); ws = ws2; }
    t_vs = time_it([&](int i) {
      if (hipsparseLtMatmul(&v.handle, &v.plan, &alpha, vblob[i], dB, &beta, dCv[i], dCv[i], dws, &st, 1) != HIPSPARSE_STATUS_SUCCESS) vendor_ok = false;
    });
  }
  // agreement of the two products (same pruned operand): max |diff| relative to max |C|
  std::vector<uint16_t> hc(M * N), hcv(M * N);
  CK(hipMemcpy(hc.data(), dC[0], M * N * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hcv.data(), dCv[0], M * N * 2, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t i = 0; i < M * N; ++i) { const double a = h2f(hc[i]), b = h2f(hcv[i]); md = std::max(md, fabs(a - b)); mx = std::max(mx, fabs(a)); }
  const double bytes = (double)s.b * (s.m * K * 2 * 9.0 / 16.0 + s.m * N * 2.0) + 2.0 * K * N;
  fprintf(out, "%6zu %5zu %5zu %3zu x%d  sm_spmma_f16 %8.1f us  hipsparseLtMatmul %8.1f us (after its search %8.1f)  ratio %.2f  roof %6.1f us  C max|diff|/max|C| %.2e%s\n",
          s.m, N, K, s.b, s.count, t_sm, t_v, t_vs, (t_vs > 0 ? std::min(t_v, t_vs) : t_v) / t_sm, bytes / 8e12 * 1e6, md / (mx > 0 ? mx : 1), vendor_ok ? "" : "  [vendor call failed]");
  fflush(out);
  t_sm_tot += t_sm * s.count; t_v_tot += t_v * s.count; t_vs_tot += (t_vs > 0 ? std::min(t_v, t_vs) : t_v) * s.count;
  for (int i = 0; i < NSET; ++i) { CK(hipFree(dA[i])); CK(hipFree(dC[i])); CK(hipFree(dCv[i])); CK(hipFree(vblob[i])); CK(hipFree(blob[i])); }
  CK(hipFree(dB)); if (dcbuf) CK(hipFree(dcbuf)); if (dws) CK(hipFree(dws));
  v.destroy();
  return vendor_ok;
}

int main(int argc, char** argv) {
  const std::string table = argc > 1 ? argv[1] : "datasets/resnet50.csv";
  FILE* out = stdout;
  if (int rc = sm_device_check()) { fprintf(stderr, "sm_device_check: %d %s\n", rc, sm_last_error()); return 2; }
  fprintf(out, "# hipSPARSELt yardstick (tools/hipsparselt_yardstick.cpp): libhipsparselt next to %s\n", sm_version());
  fprintf(out, "# (a) semantics of hipsparseLtSpMMAPrune / PruneCheck vs sm_prune24_f16 / sm_prune24_check_f16\n");
  if (!semantics(out, 3136, 512)) fprintf(out, "semantics: vendor library refused (see stderr)\n");
  if (!semantics(out, 784, 4608)) fprintf(out, "semantics: vendor library refused (see stderr)\n");
  // (b) time, unique shapes of the table with their multiplicity
  std::ifstream f(table);
  if (!f) { fprintf(stderr, "cannot open %s\n", table.c_str()); return 2; }
  std::string line; std::getline(f, line);
  std::vector<Shape> shapes;
  while (std::getline(f, line)) {
    std::stringstream ss(line); std::string t; size_t v[4]; int i = 0;
    while (std::getline(ss, t, ',') && i < 4) v[i++] = std::stoull(t);
    if (i < 4) continue;
    bool found = false;
    for (auto& s : shapes) if (s.m == v[0] && s.n == v[1] && s.k == v[2] && s.b == v[3]) { ++s.count; found = true; }
    if (!found) shapes.push_back({v[0], v[1], v[2], v[3], 1});
  }
  if (argc > 2) for (auto& s : shapes) s.b = std::stoull(argv[2]);
  fprintf(out, "# (b) 2:4 matmul on one STRIP-pruned operand, row-major, A = (b*m) x k as one tall matrix, B shared; us per launch, %s\n", table.c_str());
  double a = 0, b = 0, c = 0; int failed = 0;
  for (const auto& s : shapes) {
    if (s.k % 8 != 0) { fprintf(out, "%6zu %5zu %5zu %3zu x%d  skipped (k %% 8 != 0)\n", s.m, s.n, s.k, s.b, s.count); continue; }
    if (!time_shape(out, s, a, b, c)) ++failed;
  }
  fprintf(out, "table total (count-weighted): sm_spmma_f16 %.1f us, hipsparseLtMatmul %.1f us (best of default / searched %.1f us); %d shapes failed in the vendor library\n", a, b, c, failed);
  return 0;
}
