// spmma_i8 m n k b -- the int8 2:4 path through the C ABI (extension of this build: the vendor call behind the reference's
// include/sparsify.me/spmma.hxx:40-113 lists int8 among its 2:4 types).  Operand roles as in examples/spmma.cu:48-59 --
// b matrices A (m x k, row-major), one shared B -- except that B is given [n][k] (k-contiguous per output column; made
// here from the row-major k x n B with sm_transpose_i8) and C is int8, requantised from the int32 accumulators.
// Prints the stage times with the labels of the fp16 driver and checks that the fused kernel returns the same bytes.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include <sparsify.me/containers/vector.hxx>
#include <sparsify.me/util/util.hxx>
#include <sparsifyme.h>

int main(int argc, char** argv) {
  using namespace sparsifyme;
  if (argc != 5) {
    std::cout << "Invalid # of arguments. Usage: ./spmma_i8 m n k b" << std::endl;
    return EXIT_FAILURE;
  }
  if (sm_device_check() != SM_STATUS_SUCCESS) {
    std::cerr << "\nlibsparsifyme is supported only on gfx950 (MI355X) devices: " << sm_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  const std::size_t m = std::stoi(argv[1]), n = std::stoi(argv[2]), k = std::stoi(argv[3]), b = std::stoi(argv[4]);
  host_vector<signed char> h_A(m * k * b), h_B(k * n);
  for (auto& a : h_A) a = static_cast<signed char>(util::get_random<float>(-128.f, 127.99f));
  for (auto& x : h_B) x = static_cast<signed char>(util::get_random<float>(-128.f, 127.99f));
  device_vector<signed char> A = h_A, B = h_B, Bt(k * n), C(m * n * b), C2(m * n * b);
  int rc = sm_transpose_i8(B.data().get(), Bt.data().get(), k, n, nullptr);

  util::timer_t t;
  device_vector<int> valid(1);
  t.begin();
  rc |= sm_prune24_i8(A.data().get(), A.data().get(), m * b, k, k, SM_PRUNE_STRIP, nullptr);
  rc |= sm_prune24_check_i8(A.data().get(), m * b, k, k, valid.data().get(), nullptr);
  const float prune_ms = t.end();
  std::size_t bytes = 0;
  rc |= sm_compress24_size(m, k, 1, b, &bytes);
  device_vector<unsigned char> blob(bytes);
  t.begin();
  rc |= sm_compress24_i8(A.data().get(), m, k, k, b, m * k, blob.data().get(), nullptr);
  const float compress_ms = t.end();
  const float scale = 1.0f / (64.0f * static_cast<float>(k));
  t.begin();
  rc |= sm_spmma_i8_q(blob.data().get(), Bt.data().get(), C.data().get(), m, n, k, b, 0, m * n, scale, nullptr);
  const float mul_ms = t.end();
  if (rc != SM_STATUS_SUCCESS) {
    std::cerr << "spmma_i8: " << sm_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  std::cout << "Pruning Time (ms): " << prune_ms << std::endl;
  std::cout << "Compression Time (ms): " << compress_ms << std::endl;
  std::cout << "SpMMA Time (ms): " << mul_ms << std::endl;
  // the one-kernel form on the (already pruned, so identical) dense A
  t.begin();
  rc = sm_spmma_fused_i8_q(A.data().get(), Bt.data().get(), C2.data().get(), m, n, k, k, b, m * k, 0, m * n, scale, nullptr);
  const float fused_ms = t.end();
  if (rc == SM_STATUS_SUCCESS) {
    (void)hipDeviceSynchronize();
    const auto h1 = C.to_host(), h2 = C2.to_host();
    const bool same = std::memcmp(h1.data(), h2.data(), h1.size()) == 0;
    std::cout << "Fused Time (ms): " << fused_ms << std::endl;
    std::cout << "Fused matches: " << (same ? "yes" : "NO") << std::endl;
    if (!same) return EXIT_FAILURE;
  } else {
    std::cout << "Fused: not taken for this shape (" << sm_last_error() << ")" << std::endl;
  }
  return EXIT_SUCCESS;
}
