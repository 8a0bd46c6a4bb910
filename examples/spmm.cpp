// spmm m n k b -- Blocked-ELL x dense batch: b matrices with 2x2 blocks and ell_cols = k/2 (50 %
// block-sparse), values 1,2,3,..., per block row sorted distinct random block columns, one shared
// dense B = 1..k*n; prints the elapsed milliseconds (set-up and CLI of the reference's
// examples/spmm.cu:24-118).
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include <sparsify.me/containers/ell.hxx>
#include <sparsify.me/spmm.hxx>
#include <sparsify.me/util/util.hxx>

int main(int argc, char** argv) {
  using namespace sparsifyme;
  using type_t = float;
  if (argc != 5) {
    std::cout << "Invalid # of arguments. Usage: ./spmm m n k b" << std::endl;
    return EXIT_FAILURE;
  }
  std::size_t m = std::stoi(argv[1]), n = std::stoi(argv[2]), k = std::stoi(argv[3]), batch_size = std::stoi(argv[4]);
  const std::size_t block_size = 2;

  std::vector<ell_t<type_t, memory_space_t::device>> d_As(batch_size);
  std::mt19937 gen(0x5eed);
  for (std::size_t b = 0; b < batch_size; ++b) {
    ell_t<type_t, memory_space_t::host> h;
    h.rows = m; h.cols = k; h.block_size = block_size; h.ell_cols = k / 2;
    h.blocked_rows = m / block_size; h.blocked_cols = h.ell_cols / block_size;
    h.num_blocks = h.blocked_rows * h.blocked_cols;
    h.values.resize(h.rows * h.ell_cols);
    std::iota(h.values.begin(), h.values.end(), type_t(1));
    h.column_indices.resize(h.num_blocks);
    std::vector<std::size_t> all(k / block_size);
    std::iota(all.begin(), all.end(), std::size_t(0));
    for (std::size_t r = 0; r < h.blocked_rows; ++r) {
      std::shuffle(all.begin(), all.end(), gen);
      std::copy(all.begin(), all.begin() + h.blocked_cols, h.column_indices.begin() + r * h.blocked_cols);
      std::sort(h.column_indices.begin() + r * h.blocked_cols, h.column_indices.begin() + (r + 1) * h.blocked_cols);
    }
    d_As[b] = h;
  }
  host_vector<type_t> h_B(k * n);
  std::iota(h_B.begin(), h_B.end(), type_t(1));
  device_vector<type_t> d_B = h_B;
  std::vector<device_vector<type_t>> d_C(batch_size);
  std::vector<type_t*> Cs(batch_size);
  for (std::size_t b = 0; b < batch_size; ++b) {
    d_C[b].resize(m * n);
    Cs[b] = d_C[b].data().get();
  }
  float elapsed = batched::spmm(d_As.data(), d_B.data().get(), Cs.data(), m, n, k, batch_size);
  std::cout << elapsed << std::endl;
  return EXIT_SUCCESS;
}
