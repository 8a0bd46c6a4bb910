// tests/cpp/header_parity.cpp -- TEST INFRASTRUCTURE: values computed THROUGH THE C++ HEADERS (include/sparsify.me/*.hxx,
// the drop-in boundary the reference's drivers are written against: examples/{sparsify,gemm,spmm,batched_coo,spmma}.cu)
// compared with the CPU oracle (oracle/libsm_oracle.so).  The C-ABI parity tests cannot see a wrong pointer table, a
// swapped argument or a dropped transpose flag in a header template; this binary can.  It links the product library
// (through the headers) AND the oracle, which only tests may do.
//
//   header_parity table.csv            every check on every row of the table; exit 0 iff all pass
//   header_parity table.csv --swap     self-test of the comparator: the C pointer table handed to batched::spmm /
//                                      batched::gemm is rotated by one batch while the expectation is not; every such
//                                      check must FAIL (exit 0 iff they all do) -- "a swapped pointer turns the suite red"
//
// Tolerances: bit-exact for the positional sparsify, the pruned A and its validity; GEMM-type outputs against fp64
// accumulation within ROUND * |ref| + 2k * 2^-24 * sum|a b| (ROUND = 2^-22 fp32, 2^-10 fp16), the bound
// tests/test_gpu_parity.py uses.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include <sparsify.me/containers/ell.hxx>
#include <sparsify.me/gemm.hxx>
#include <sparsify.me/sparsify.hxx>
#include <sparsify.me/spmm.hxx>
#include <sparsify.me/spmma.hxx>
#include <sparsify.me/util/util.hxx>

using namespace sparsifyme;

// ---- the oracle's C entry points (oracle/sm_oracle.c)
extern "C" {
int sm_sparsify_positional_ref(void* weights, uint64_t* mask, size_t m, size_t n, size_t elt_bytes, size_t blk_m, size_t blk_n, float sf);
int sm_prune24_f16_ref(const uint16_t* A_in, uint16_t* A_out, size_t m, size_t k, size_t ld, int alg);
int sm_prune24_f32_ref(const float* A_in, float* A_out, size_t m, size_t k, size_t ld, int alg);
int sm_compress24_size_ref(size_t m, size_t k, size_t elt_bytes, size_t batch, size_t* bytes);
int sm_compress24_f16_ref(const uint16_t* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob);
int sm_compress24_f32_ref(const float* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob);
int sm_spmma_f16_ref(const void* blob, const uint16_t* B, uint16_t* C, size_t m, size_t n, size_t k, size_t batch, size_t strideB, size_t strideC, float alpha, float beta);
int sm_spmma_f32_ref(const void* blob, const float* B, float* C, size_t m, size_t n, size_t k, size_t batch, size_t strideB, size_t strideC, float alpha, float beta);
int sm_gemm_batched_f32_ref(void* const* A, void* const* B, void* const* C, size_t m, size_t n, size_t k, size_t batch, int ta, int tb, float alpha, float beta);
int sm_gemm_batched_f16_ref(void* const* A, void* const* B, void* const* C, size_t m, size_t n, size_t k, size_t batch, int ta, int tb, float alpha, float beta);
int sm_spmm_bell_f32_ref(const float* values, const uint64_t* column_indices, size_t rows, size_t cols, size_t block_size, size_t ell_cols, const float* B, float* C, size_t n, float alpha, float beta);
int sm_spmm_coo_f32_ref(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches, const int* rows, const int* cols, const float* vals, const float* B, float* C, float alpha, float beta);
}

static int g_fail = 0, g_checks = 0;
static bool g_swap = false;

static float h2f(uint16_t h) { __half x; std::memcpy(&x, &h, 2); return __half2float(x); }
static uint16_t f2h(float f) { __half x = __float2half(f); uint16_t h; std::memcpy(&h, &x, 2); return h; }

// `expect_fail`: a comparator self-test (--swap): the check counts as passed iff the values DIFFER
static void verdict(const char* what, bool equal, bool expect_fail, const std::string& detail) {
  ++g_checks;
  const bool ok = expect_fail ? !equal : equal;
  if (!ok) ++g_fail;
  std::printf("%-58s %s%s%s\n", what, ok ? "ok" : "MISMATCH", expect_fail ? (equal ? "  (comparator did not notice the rotated pointer table)" : "  (rotated pointer table detected)") : "",
              detail.empty() ? "" : ("  " + detail).c_str());
}

// |got - ref| <= round * |ref| + 2 k 2^-24 scale, element by element
static bool close_enough(const std::vector<float>& got, const std::vector<float>& ref, const std::vector<double>& scale, double round, size_t k, std::string& detail) {
  double worst = 0;
  size_t at = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    const double tol = round * std::fabs((double)ref[i]) + 2.0 * k * std::ldexp(1.0, -24) * scale[i] + 1e-30;
    const double r = std::fabs((double)got[i] - (double)ref[i]) / tol;
    if (!(r <= worst)) { worst = r; at = i; }  // NaN-safe: a NaN ratio becomes the worst
  }
  char buf[128];
  std::snprintf(buf, sizeof(buf), "worst |diff| / bound = %.3f at %zu", worst, at);
  detail = buf;
  return worst <= 1.0;
}

template <typename T>
static std::vector<T> to_host(const device_vector<T>& d) { return d.to_host(); }

// ------------------------------------------------------------------------------------------------------------------
static void check_sparsify(size_t m, size_t n) {
  for (float sf : {0.5f, 0.25f, 1.0f}) {
    host_vector<float> w(m * n);
    std::iota(w.begin(), w.end(), 1.0f);
    device_vector<float> dw = w;
    device_vector<std::size_t> dmask(m * n);
    sparsify<2, 2>(dw.data().get(), dmask.data().get(), m, n, sf);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> mask_ref(m * n, 7);
    sm_sparsify_positional_ref(w.data(), mask_ref.data(), m, n, 4, 2, 2, sf);
    const auto gw = to_host(dw);
    const auto gm = to_host(dmask);
    bool eq = !std::memcmp(gw.data(), w.data(), m * n * 4);
    for (size_t i = 0; i < m * n && eq; ++i) eq = gm[i] == mask_ref[i];
    char what[96];
    std::snprintf(what, sizeof(what), "sparsify<2,2,float> %zux%zu sf=%.2f (weights + mask, bit-exact)", m, n, sf);
    verdict(what, eq, false, "");
  }
}

static void check_gemm(size_t m, size_t n, size_t k, size_t b) {
  std::mt19937 gen(0x5eed + (unsigned)(m + n + k));
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (int mode = 0; mode < 2; ++mode) {  // (N, N) and (T, N)
    // the reference always passes lda = m (gemm.hxx:80-81), so a transposed A (k x m stored, leading dimension m) only
    // exists for m >= k; for the other rows of the table the call is an argument error in the vendor library and here
    if (mode && m < k) continue;
    const size_t a_elems = mode ? m * m : m * k;
    const operation_t ta = mode ? operation_t::T : operation_t::N, tb = operation_t::N;
    std::vector<host_vector<float>> hA(b), hC(b);
    host_vector<float> hB(k * n);
    for (auto& x : hB) x = U(gen);
    std::vector<device_vector<float>> dA(b), dC(b);
    device_vector<float> dB = hB;
    std::vector<float*> pA(b), pB(b), pC(b);
    for (size_t i = 0; i < b; ++i) {
      hA[i].resize(a_elems);
      for (auto& x : hA[i]) x = U(gen);
      dA[i] = hA[i];
      dC[i].resize(m * n);
      pA[i] = dA[i].data().get(); pB[i] = dB.data().get(); pC[i] = dC[i].data().get();
    }
    std::vector<float*> pC_call = pC;
    if (g_swap && b > 1) std::rotate(pC_call.begin(), pC_call.begin() + 1, pC_call.end());
    device_vector<float*> dpA = pA, dpB = pB, dpC = pC_call;  // device arrays of device pointers (examples/gemm.cu:65-90)
    batched::gemm<float>(dpA.data().get(), dpB.data().get(), dpC.data().get(), m, n, k, b, ta, tb);
    (void)hipDeviceSynchronize();
    // oracle: column-major, lda = m (op(A)(r, l) = A[r * m + l] when transposed, A[l * m + r] otherwise), ldb = k, ldc = m
    bool all = true;
    std::string detail;
    for (size_t i = 0; i < b; ++i) {
      std::vector<float> ref(m * n, 0.f);
      void* a1 = hA[i].data(); void* b1 = hB.data(); void* c1 = ref.data();
      sm_gemm_batched_f32_ref(&a1, &b1, &c1, m, n, k, 1, (int)ta, (int)tb, 1.f, 0.f);
      std::vector<double> scale(m * n, 0.0);
      for (size_t j = 0; j < n; ++j)
        for (size_t r = 0; r < m; ++r) {
          double s = 0;
          for (size_t l = 0; l < k; ++l) s += std::fabs((double)(mode ? hA[i][r * m + l] : hA[i][l * m + r])) * std::fabs((double)hB[j * k + l]);
          scale[j * m + r] = s;
        }
      all = close_enough(to_host(dC[i]), ref, scale, std::ldexp(1.0, -22), k, detail) && all;
    }
    char what[96];
    std::snprintf(what, sizeof(what), "batched::gemm<float> %zux%zux%zu b=%zu (%s,N) vs oracle", m, n, k, b, mode ? "T" : "N");
    verdict(what, all, g_swap && b > 1, detail);
  }
}

static void check_spmm(size_t m, size_t n, size_t k, size_t b) {
  const size_t bs = 2;
  std::mt19937 gen(0xbe11 + (unsigned)(m * 3 + k));
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<ell_t<float, memory_space_t::host>> hAs(b);
  std::vector<ell_t<float, memory_space_t::device>> dAs(b);
  for (size_t i = 0; i < b; ++i) {
    auto& h = hAs[i];
    h.rows = m; h.cols = k; h.block_size = bs; h.ell_cols = k / 2;
    h.blocked_rows = m / bs; h.blocked_cols = h.ell_cols / bs; h.num_blocks = h.blocked_rows * h.blocked_cols;
    h.values.resize(h.rows * h.ell_cols);
    for (auto& x : h.values) x = U(gen);
    h.column_indices.resize(h.num_blocks);
    std::vector<std::size_t> all(k / bs);
    std::iota(all.begin(), all.end(), std::size_t(0));
    for (size_t r = 0; r < h.blocked_rows; ++r) {
      std::shuffle(all.begin(), all.end(), gen);
      std::copy(all.begin(), all.begin() + h.blocked_cols, h.column_indices.begin() + r * h.blocked_cols);
      std::sort(h.column_indices.begin() + r * h.blocked_cols, h.column_indices.begin() + (r + 1) * h.blocked_cols);
    }
    dAs[i] = h;
  }
  host_vector<float> hB(k * n);
  for (auto& x : hB) x = U(gen);
  device_vector<float> dB = hB;
  std::vector<device_vector<float>> dC(b);
  std::vector<float*> Cs(b);
  for (size_t i = 0; i < b; ++i) { dC[i].resize(m * n); Cs[i] = dC[i].data().get(); }
  std::vector<float*> Cs_call = Cs;
  if (g_swap && b > 1) std::rotate(Cs_call.begin(), Cs_call.begin() + 1, Cs_call.end());
  batched::spmm<float>(dAs.data(), dB.data().get(), Cs_call.data(), m, n, k, b);
  (void)hipDeviceSynchronize();
  bool all = true;
  std::string detail;
  for (size_t i = 0; i < b; ++i) {
    std::vector<float> ref(m * n, 0.f);
    std::vector<uint64_t> ci(hAs[i].column_indices.begin(), hAs[i].column_indices.end());
    sm_spmm_bell_f32_ref(hAs[i].values.data(), ci.data(), m, k, bs, k / 2, hB.data(), ref.data(), n, 1.f, 0.f);
    std::vector<double> scale(m * n, 0.0);
    const size_t bcols = (k / 2) / bs;
    for (size_t j = 0; j < n; ++j)
      for (size_t r = 0; r < m; ++r) {
        double s = 0;
        for (size_t e = 0; e < bcols; ++e)
          for (size_t t = 0; t < bs; ++t)
            s += std::fabs((double)hAs[i].values[r * (k / 2) + e * bs + t]) * std::fabs((double)hB[j * k + ci[(r / bs) * bcols + e] * bs + t]);
        scale[j * m + r] = s;
      }
    all = close_enough(to_host(dC[i]), ref, scale, std::ldexp(1.0, -22), k, detail) && all;
  }
  char what[96];
  std::snprintf(what, sizeof(what), "batched::spmm<float> (Blocked-ELL 2x2) %zux%zux%zu b=%zu vs oracle", m, n, k, b);
  verdict(what, all, g_swap && b > 1, detail);
}

static void check_coo(size_t m, size_t n, size_t k, size_t b) {
  std::mt19937 gen(0xc00 + (unsigned)(m + 7 * k));
  std::uniform_real_distribution<float> U(-1.f, 1.f), P(0.f, 1.f);
  std::vector<int> rows, cols;
  std::vector<float> vals;
  for (size_t r = 0; r < m; ++r)
    for (size_t c = 0; c < k; ++c)
      if (P(gen) < 0.1f) { rows.push_back((int)r); cols.push_back((int)c); vals.push_back(U(gen)); }
  const size_t nnz = vals.size();
  host_vector<float> hB(b * k * n);
  for (auto& x : hB) x = U(gen);
  device_vector<int> dr = rows, dc = cols;
  device_vector<float> dv = vals, dB = hB, dC(b * m * n);
  std::vector<float> ref(b * m * n, 0.f);
  sm_spmm_coo_f32_ref(m, k, nnz, n, b, rows.data(), cols.data(), vals.data(), hB.data(), ref.data(), 1.f, 0.f);
  std::vector<double> scale(b * m * n, 0.0);
  for (size_t bi = 0; bi < b; ++bi)
    for (size_t e = 0; e < nnz; ++e)
      for (size_t j = 0; j < n; ++j)
        scale[bi * m * n + j * m + rows[e]] += std::fabs((double)vals[e]) * std::fabs((double)hB[bi * k * n + j * k + cols[e]]);
  // default: the fp32 kernels, held to fp32 arithmetic; strided_coo_options().fast = true: the fp16-split dense-MFMA form first where
  // the library takes the shape -- then the bound is one fp16 rounding of the dense operand, 2^-11 of sum|a||b| (include/sparsifyme.h)
  for (int exact = 1; exact >= 0; --exact) {
    batched::strided_coo_options().fast = exact == 0;
    (void)hipMemset(dC.data().get(), 0xff, b * m * n * sizeof(float));
    batched::strided_coo<float>(m, k, nnz, k, n, b, dr.data().get(), dc.data().get(), dv.data().get(), dB.data().get(), dC.data().get());
    (void)hipDeviceSynchronize();
    std::string detail;
    std::vector<double> sc = scale;
    if (!exact)
      for (auto& x : sc) x *= 1.0 + 1.02 * std::ldexp(1.0, -11) / (2.0 * k * std::ldexp(1.0, -24));  // + 1.02 * 2^-11 * scale on top of the fp32 terms
    const bool ok = close_enough(to_host(dC), ref, sc, std::ldexp(1.0, -22), k, detail);
    char what[128];
    std::snprintf(what, sizeof(what), "batched::strided_coo<float> (%s) %zux%zux%zu b=%zu nnz=%zu vs oracle", exact ? "default: exact fp32" : "opt-in fast form", m, n, k, b, nnz);
    verdict(what, ok, false, detail);
  }
  batched::strided_coo_options().fast = false;
}

// spmma<type_t>(): TILE prune in place + check + compress + multiply; (N, N) and (T, N): the in-place pruned A must be the
// oracle's TILE-pruned A bit for bit (in its stored orientation), C the oracle's product of the compressed operand
// planes = 3 / 2 (float only): spmma_options().f32_planes -- the multiply on the sparse matrix instruction through bfloat16 splits;
// same pruned A bit for bit, C within 2^-21 / 2^-13 of sum|a||b| on top of the fp32 terms (include/sparsifyme.h)
template <typename T>
static void check_spmma(size_t m, size_t n, size_t k, size_t b, const char* tname, int planes = 0, bool fewest = false) {
  constexpr bool F32 = sizeof(T) == 4;
  spmma_options().f32_planes = planes;
  spmma_options().fewest_passes = fewest;  // (round 6) the whole sequence through sm_prune24_spmma_*: same dA, same dC
  std::mt19937 gen(0xa24 + (unsigned)(m + n * 3 + k * 5));
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  using bits_t = typename std::conditional<F32, float, uint16_t>::type;
  auto enc = [](float x) -> bits_t { if constexpr (F32) return x; else return f2h(x); };
  auto dec = [](bits_t x) -> float { if constexpr (F32) return x; else return h2f(x); };
  for (int mode = 0; mode < 2; ++mode) {
    const bool ta = mode == 1;
    std::vector<bits_t> hA(b * m * k), hB(b * k * n);  // hA in the N form (m x k row-major per batch); B per batch (examples/spmma.cu:48-59)
    for (auto& x : hA) x = enc(U(gen));
    for (auto& x : hB) x = enc(U(gen));
    std::vector<bits_t> stored = hA;  // what the caller holds: k x m row-major per batch when transpose_a
    if (ta)
      for (size_t bi = 0; bi < b; ++bi)
        for (size_t r = 0; r < m; ++r)
          for (size_t c = 0; c < k; ++c) stored[bi * m * k + c * m + r] = hA[bi * m * k + r * k + c];
    device_vector<bits_t> dA = stored, dB = hB, dC(b * m * n);
    spmma<T>(reinterpret_cast<T*>(dA.data().get()), reinterpret_cast<T*>(dB.data().get()), reinterpret_cast<T*>(dC.data().get()), m, n, k, b,
             ta ? operation_t::T : operation_t::N, operation_t::N);
    (void)hipDeviceSynchronize();
    // oracle: TILE prune per batch when a 4 x 4 tile would straddle batches, else as one tall matrix (what the header does)
    std::vector<bits_t> pr(b * m * k);
    if (m % 4 == 0 || b == 1) {
      if constexpr (F32) sm_prune24_f32_ref(hA.data(), pr.data(), m * b, k, k, 0); else sm_prune24_f16_ref(hA.data(), pr.data(), m * b, k, k, 0);
    } else {
      for (size_t bi = 0; bi < b; ++bi) {
        if constexpr (F32) sm_prune24_f32_ref(hA.data() + bi * m * k, pr.data() + bi * m * k, m, k, k, 0);
        else sm_prune24_f16_ref(hA.data() + bi * m * k, pr.data() + bi * m * k, m, k, k, 0);
      }
    }
    std::vector<bits_t> pr_stored = pr;
    if (ta)
      for (size_t bi = 0; bi < b; ++bi)
        for (size_t r = 0; r < m; ++r)
          for (size_t c = 0; c < k; ++c) pr_stored[bi * m * k + c * m + r] = pr[bi * m * k + r * k + c];
    const auto gotA = to_host(dA);
    char what[112];
    std::snprintf(what, sizeof(what), "spmma<%s> %zux%zux%zu b=%zu (%s,N): pruned A in place (TILE, bit-exact)", tname, m, n, k, b, ta ? "T" : "N");
    verdict(what, !std::memcmp(gotA.data(), pr_stored.data(), gotA.size() * sizeof(bits_t)), false, "");
    size_t bytes = 0;
    sm_compress24_size_ref(m, k, sizeof(bits_t), b, &bytes);
    std::vector<unsigned char> blob(bytes);
    std::vector<bits_t> cref(b * m * n);
    if constexpr (F32) { sm_compress24_f32_ref(pr.data(), m, k, k, b, m * k, blob.data()); sm_spmma_f32_ref(blob.data(), hB.data(), cref.data(), m, n, k, b, k * n, m * n, 1.f, 0.f); }
    else { sm_compress24_f16_ref(pr.data(), m, k, k, b, m * k, blob.data()); sm_spmma_f16_ref(blob.data(), hB.data(), cref.data(), m, n, k, b, k * n, m * n, 1.f, 0.f); }
    std::vector<float> got(b * m * n), ref(b * m * n);
    std::vector<double> scale(b * m * n, 0.0);
    const auto gotC = to_host(dC);
    for (size_t i = 0; i < got.size(); ++i) { got[i] = dec(gotC[i]); ref[i] = dec(cref[i]); }
    for (size_t bi = 0; bi < b; ++bi)
      for (size_t r = 0; r < m; ++r)
        for (size_t l = 0; l < k; ++l) {
          const double a = std::fabs((double)dec(pr[bi * m * k + r * k + l]));
          if (a == 0.0) continue;
          for (size_t j = 0; j < n; ++j) scale[bi * m * n + r * n + j] += a * std::fabs((double)dec(hB[bi * k * n + l * n + j]));
        }
    std::string detail;
    if (planes)
      for (auto& x : scale) x *= 1.0 + std::ldexp(1.0, planes == 3 ? -21 : -13) / (2.0 * k * std::ldexp(1.0, -24));
    // the oracle's C is already rounded to the type: allow one more rounding of the output on top of the accumulation bound
    const bool ok = close_enough(got, ref, scale, F32 ? std::ldexp(1.0, -22) : std::ldexp(1.0, -9), k, detail);
    std::snprintf(what, sizeof(what), "spmma<%s> %zux%zux%zu b=%zu (%s,N): C vs oracle", tname, m, n, k, b, ta ? "T" : "N");
    verdict(what, ok, false, detail);
  }
  spmma_options().f32_planes = 0;
  spmma_options().fewest_passes = false;
}

// spmma_f32_planes_t (round 5): B's planes prepared once == spmma_fused<float> with the same f32_planes, bit for bit, over two multiplies
static void check_prepared_planes(size_t m, size_t n, size_t k, size_t b, int planes) {
  if (k % 64 != 0 || n % 8 != 0) return;  // shapes the split form takes without the span rules
  std::mt19937 gen(0x9a1 + (unsigned)(m + n * 3 + k * 5));
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> hA(b * m * k), hB(b * k * n);  // B per batch, as spmma_fused takes it (examples/spmma.cu:48-59)
  for (auto& x : hA) x = U(gen);
  for (auto& x : hB) x = U(gen);
  device_vector<float> dA = hA, dB = hB, dC0(b * m * n), dC1(b * m * n);
  spmma_options().f32_planes = planes;
  spmma_fused<float>(dA.data().get(), dB.data().get(), dC0.data().get(), m, n, k, b);
  spmma_options().f32_planes = 0;
  spmma_f32_planes_t pl(k, n, planes, b, true);
  bool ok = pl.prepare(dB.data().get()) == SM_STATUS_SUCCESS;
  std::vector<float> h0(b * m * n), h1(b * m * n);
  (void)hipMemcpy(h0.data(), dC0.data().get(), h0.size() * 4, hipMemcpyDeviceToHost);
  for (int rep = 0; rep < 2 && ok; ++rep) {
    (void)hipMemset(dC1.data().get(), 0xff, h1.size() * 4);
    ok = pl.multiply(dA.data().get(), dC1.data().get(), m) == SM_STATUS_SUCCESS;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h1.data(), dC1.data().get(), h1.size() * 4, hipMemcpyDeviceToHost);
    ok = ok && std::memcmp(h0.data(), h1.data(), h0.size() * 4) == 0;
  }
  char what[200];
  std::snprintf(what, sizeof(what), "spmma_f32_planes_t planes = %d %zux%zux%zu b=%zu: C == spmma_fused<float> bit for bit", planes, m, n, k, b);
  verdict(what, ok, false, "");
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cout << "Usage: ./header_parity table.csv [--swap]" << std::endl;
    return EXIT_FAILURE;
  }
  g_swap = argc > 2 && std::string(argv[2]) == "--swap";
  std::vector<util::mat_sz> shapes;
  try {
    shapes = util::read_shapes(argv[1]);
  } catch (const char* e) {
    std::cerr << e << std::endl;
    return EXIT_FAILURE;
  }
  for (const auto& s : shapes) {
    const size_t m = std::get<0>(s), n = std::get<1>(s), k = std::get<2>(s), b = std::get<3>(s);
    if (g_swap) {  // only the checks that take a pointer table
      check_gemm(m, n, k, b);
      check_spmm(m, n, k, b);
      continue;
    }
    check_sparsify(m, k);
    check_gemm(m, n, k, b);
    check_spmm(m, n, k, b);
    check_coo(m, n, k, b);
    check_spmma<_Float16>(m, n, k, b, "half");
    check_spmma<_Float16>(m, n, k, b, "half, fewest_passes", 0, true);
    check_spmma<float>(m, n, k, b, "float");
    check_spmma<float>(m, n, k, b, "float, f32_planes = 3", 3);
    check_spmma<float>(m, n, k, b, "float, f32_planes = 2", 2);
    check_prepared_planes(m, n, k, b, 3);
    check_prepared_planes(m, n, k, b, 2);
  }
  std::printf("%d checks, %d failed%s\n", g_checks, g_fail, g_swap ? " (--swap: a check passes iff the rotated table was noticed)" : "");
  return g_fail == 0 && g_checks > 0 ? EXIT_SUCCESS : EXIT_FAILURE;
}
