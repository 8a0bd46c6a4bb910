// gemm m n k b -- times sparsifyme::batched::gemm: b distinct A (m x k), ONE shared B (k x n) whose
// pointer is repeated b times, b outputs; column-major; prints the elapsed milliseconds (the CLI,
// operand set-up and output of the reference's examples/gemm.cu:21-97).
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include <sparsify.me/containers/vector.hxx>
#include <sparsify.me/gemm.hxx>
#include <sparsify.me/util/util.hxx>

#ifndef SM_TYPE
#define SM_TYPE float
#endif

int main(int argc, char** argv) {
  using namespace sparsifyme;
  using type_t = SM_TYPE;
  if (argc != 5) {
    std::cout << "Invalid # of input args. Usage: ./gemm m n k b" << std::endl;
    return EXIT_FAILURE;
  }
  std::size_t m = std::stoi(argv[1]), n = std::stoi(argv[2]), k = std::stoi(argv[3]), batch_size = std::stoi(argv[4]);

  host_vector<type_t> h_B(k * n);
  for (auto& v : h_B) v = static_cast<type_t>(util::get_random<float>());
  device_vector<type_t> d_B = h_B;

  std::vector<device_vector<type_t>> d_A(batch_size), d_C(batch_size);
  host_vector<type_t*> hA(batch_size), hB(batch_size), hC(batch_size);
  host_vector<type_t> h_A(m * k);
  for (std::size_t b = 0; b < batch_size; ++b) {
    for (auto& v : h_A) v = static_cast<type_t>(util::get_random<float>());
    d_A[b] = h_A;
    d_C[b].resize(m * n);
    hA[b] = d_A[b].data().get();
    hB[b] = d_B.data().get();
    hC[b] = d_C[b].data().get();
  }
  device_vector<type_t*> pA = hA, pB = hB, pC = hC;  // device arrays of device pointers

  float elapsed = batched::gemm(pA.data().get(), pB.data().get(), pC.data().get(), m, n, k, batch_size);
  std::cout << elapsed << std::endl;
  return EXIT_SUCCESS;
}
