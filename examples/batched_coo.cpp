// batched_coo m n k b -- one COO matrix A (m x k, density 0.1, values U[-1,1)) times a strided
// batch of b dense matrices B_i (k x n); prints the elapsed milliseconds.  Follows the INTENT of the
// reference's examples/batched_coo.cu:31-112, which is broken in several ways as committed (index
// arrays sized by rows/cols instead of nnz, :56,64; memcpy sizes in elements, :89,93; nnz from
// m*n*0.5, :46); the density is the 0.1 of profiling/python/gemm_coo_compare.py:7.
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include <sparsify.me/containers/vector.hxx>
#include <sparsify.me/spmm.hxx>
#include <sparsify.me/util/gen.hxx>

int main(int argc, char** argv) {
  using namespace sparsifyme;
  if (argc != 5) {
    std::cout << "Invalid # of arguments. Usage: ./batched_coo m n k b" << std::endl;
    return EXIT_FAILURE;
  }
  std::size_t m = std::stoi(argv[1]), n = std::stoi(argv[2]), k = std::stoi(argv[3]), b = std::stoi(argv[4]);
  const double density = 0.1;
  std::mt19937 gen(0x5eed);
  std::uniform_real_distribution<float> val(-1.0f, 1.0f), coin(0.0f, 1.0f);
  host_vector<int> rows, cols;
  host_vector<float> vals;
  for (std::size_t i = 0; i < m; ++i)
    for (std::size_t j = 0; j < k; ++j)
      if (coin(gen) < density) {
        rows.push_back((int)i);
        cols.push_back((int)j);
        vals.push_back(val(gen));
      }
  device_vector<int> d_rows = rows, d_cols = cols;
  device_vector<float> d_vals = vals, d_B(k * n * b), d_C(m * n * b);
  util::random::uniform_distribution(d_B, -1.0f, 1.0f, 7);
  float elapsed = batched::strided_coo<float>(m, k, vals.size(), k, n, b, d_rows.data().get(), d_cols.data().get(),
                                              d_vals.data().get(), d_B.data().get(), d_C.data().get());
  std::cout << elapsed << std::endl;
  return EXIT_SUCCESS;
}
