// sweep <shapes.csv> [out.csv] -- the layer sweep in C++ through the header-only API: for every row
// (m,n,k,b) of a shape table it runs batched::gemm (column-major, one shared B), sparsify<2,2> on the
// m x k operand, batched::spmm (Blocked-ELL, 2 x 2 blocks, ell_cols = k/2, as bin/spmm builds it) and spmma, and
// writes one CSV row.  This is the in-process counterpart of the reference's examples/profiling.py:4-44 (which
// shells out to bin/gemm, bin/sparsify, bin/spmm per row and collects `m,n,k,b,gemm,prune,spmm` into compare.csv):
// the reference's seven columns come first, unchanged in name and meaning (milliseconds), followed by the three 2:4
// stage times and effective GF/s (= 2*m*n*k*b / time).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include <sparsify.me/containers/vector.hxx>
#include <sparsify.me/gemm.hxx>
#include <sparsify.me/containers/ell.hxx>
#include <sparsify.me/sparsify.hxx>
#include <sparsify.me/spmm.hxx>
#include <sparsify.me/spmma.hxx>
#include <sparsify.me/util/gen.hxx>
#include <sparsify.me/util/util.hxx>

int main(int argc, char** argv) {
  using namespace sparsifyme;
  using type_t = _Float16;
  if (argc < 2) {
    std::cout << "Usage: ./sweep shapes.csv [out.csv]" << std::endl;
    return EXIT_FAILURE;
  }
  std::vector<util::mat_sz> shapes;
  try {
    shapes = util::read_shapes(argv[1]);
  } catch (const char* msg) {
    std::cerr << msg << std::endl;
    return EXIT_FAILURE;
  }
  std::ofstream out(argc > 2 ? argv[2] : "compare.csv");
  // spmma_prune / _compress / _mul: the three separately launched and timed stages (spmma_options().staged, as the reference's
  // driver prints them); spmma_call: one spmma() call as callers get it (the fewest launches the library has for the shape)
  out << "m,n,k,b,gemm,prune,spmm,spmma_prune,spmma_compress,spmma_mul,gemm_gfs,spmma_mul_gfs,spmma_call\n";
  double tg = 0, tm = 0, ts = 0, flops = 0;
  std::mt19937 gen(0x5eed);
  for (std::size_t li = 0; li < shapes.size(); ++li) {
    const std::size_t m = std::get<0>(shapes[li]), n = std::get<1>(shapes[li]), k = std::get<2>(shapes[li]),
                      b = std::get<3>(shapes[li]);
    device_vector<type_t> A(m * k * b), B(k * n * b), C(m * n * b);
    util::random::uniform_distribution(A, 0.0f, 1.0f, 1000 + li);
    util::random::uniform_distribution(B, 0.0f, 1.0f, 2000 + li);
    host_vector<type_t*> hA(b), hB(b), hC(b);
    for (std::size_t i = 0; i < b; ++i) {
      hA[i] = A.data().get() + i * m * k;
      hB[i] = B.data().get();  // one shared B (examples/gemm.cu:60,86)
      hC[i] = C.data().get() + i * m * n;
    }
    device_vector<type_t*> pA = hA, pB = hB, pC = hC;
    batched::gemm(pA.data().get(), pB.data().get(), pC.data().get(), m, n, k, b);  // warm
    const float gemm_ms = batched::gemm(pA.data().get(), pB.data().get(), pC.data().get(), m, n, k, b);

    device_vector<type_t> W(m * k);
    device_vector<std::size_t> mask(m * k);
    util::timer_t t;
    sparsify<2, 2>(W.data().get(), mask.data().get(), m, k);
    t.begin();
    sparsify<2, 2>(W.data().get(), mask.data().get(), m, k);
    const float prune_ms = t.end();

    // batched::spmm as examples/spmm.cu:45-118 sets it up: b Blocked-ELL matrices (2 x 2 blocks, half the block
    // columns present, sorted distinct per block row), one shared dense B, fp32
    float spmm_ms = 0.0f;
    {
      const std::size_t bs = 2;
      std::vector<ell_t<float, memory_space_t::device>> As(b);
      ell_t<float, memory_space_t::host> h;
      h.rows = m; h.cols = k; h.block_size = bs; h.ell_cols = k / 2;
      h.blocked_rows = m / bs; h.blocked_cols = h.ell_cols / bs;
      h.num_blocks = h.blocked_rows * h.blocked_cols;
      h.values.resize(h.rows * h.ell_cols);
      std::iota(h.values.begin(), h.values.end(), 1.0f);
      h.column_indices.resize(h.num_blocks);
      std::vector<std::size_t> all(k / bs);
      std::iota(all.begin(), all.end(), std::size_t(0));
      for (std::size_t r = 0; r < h.blocked_rows; ++r) {
        std::shuffle(all.begin(), all.end(), gen);
        std::copy(all.begin(), all.begin() + h.blocked_cols, h.column_indices.begin() + r * h.blocked_cols);
        std::sort(h.column_indices.begin() + r * h.blocked_cols, h.column_indices.begin() + (r + 1) * h.blocked_cols);
      }
      for (std::size_t i = 0; i < b; ++i) As[i] = h;  // the same pattern per batch keeps the set-up cheap; the values differ in the reference only by position too
      device_vector<float> Bf(k * n);
      util::random::uniform_distribution(Bf, 0.0f, 1.0f, 3000 + li);
      std::vector<device_vector<float>> Cf(b);
      std::vector<float*> Cs(b);
      for (std::size_t i = 0; i < b; ++i) {
        Cf[i].resize(m * n);
        Cs[i] = Cf[i].data().get();
      }
      // (2 x 2 blocks need even m and k and whole block columns: the 7 x 7 x 3 stem layer, k = 147, has no such operand -- the reference's
      //  examples/spmm.cu:45-56 would build a truncated one; its column stays 0 here instead of an error line from the operator)
      if (h.blocked_cols > 0 && h.blocked_rows > 0 && m % bs == 0 && k % bs == 0 && h.ell_cols % bs == 0) {
        batched::spmm(As.data(), Bf.data().get(), Cs.data(), m, n, k, b);  // warm
        spmm_ms = batched::spmm(As.data(), Bf.data().get(), Cs.data(), m, n, k, b);
      }
    }

    spmma_options().staged = true;
    spmma(A.data().get(), B.data().get(), C.data().get(), m, n, k, b);  // warm (prunes A in place)
    const auto st = spmma(A.data().get(), B.data().get(), C.data().get(), m, n, k, b);
    spmma_options().staged = false;
    spmma_options().fewest_passes = true;  // spmma_call: the whole sequence through sm_prune24_spmma_* (no blob; one measured time)
    spmma(A.data().get(), B.data().get(), C.data().get(), m, n, k, b);  // warm
    const auto sc = spmma(A.data().get(), B.data().get(), C.data().get(), m, n, k, b);
    spmma_options().fewest_passes = false;
    const double fl = 2.0 * m * n * k * b;
    out << m << "," << n << "," << k << "," << b << "," << gemm_ms << "," << prune_ms << "," << spmm_ms << "," << st[0] << ","
        << st[1] << "," << st[2] << "," << fl / gemm_ms / 1e6 << "," << fl / st[2] / 1e6 << "," << sc[0] + sc[1] + sc[2] << "\n";
    tg += gemm_ms; tm += st[2]; ts += spmm_ms; flops += fl;
  }
  std::cout << "layers " << shapes.size() << "  gemm " << tg << " ms (" << flops / tg / 1e6 << " GF/s)  spmm (Blocked-ELL) " << ts
            << " ms  spmma matmul " << tm
            << " ms (" << flops / tm / 1e6 << " GF/s effective)  ratio " << tg / tm << std::endl;
  return EXIT_SUCCESS;
}
