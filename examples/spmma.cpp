// spmma m n k b -- prune to 2:4, compress, multiply; prints the three stage times with the labels
// of the reference's examples/spmma.cu:64-66.  The reference refuses every GPU that is not compute
// capability 8.0 (:35-41); this driver refuses every GPU that is not gfx950.  Element type: fp16,
// the type the reference's spmma actually declares to its back end (spmma.hxx:40); build with
// -DSM_TYPE=float for the fp32 kernels.
#include <cstdlib>
#include <iostream>
#include <string>

#include <sparsify.me/containers/vector.hxx>
#include <sparsify.me/spmma.hxx>
#include <sparsify.me/util/util.hxx>

#ifndef SM_TYPE
#define SM_TYPE _Float16
#endif

int main(int argc, char** argv) {
  using namespace sparsifyme;
  using type_t = SM_TYPE;
  if (argc != 5) {
    std::cout << "Invalid # of arguments. Usage: ./spmma m n k b" << std::endl;
    return EXIT_FAILURE;
  }
  if (sm_device_check() != SM_STATUS_SUCCESS) {
    std::cerr << "\nlibsparsifyme is supported only on gfx950 (MI355X) devices: " << sm_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  std::size_t m = std::stoi(argv[1]), n = std::stoi(argv[2]), k = std::stoi(argv[3]), batch_size = std::stoi(argv[4]);

  host_vector<type_t> h_A(m * k * batch_size), h_B(k * n * batch_size);
  for (auto& a : h_A) a = static_cast<type_t>(util::get_random<float>());
  for (auto& b : h_B) b = static_cast<type_t>(util::get_random<float>());
  device_vector<type_t> A = h_A, B = h_B, C(m * n * batch_size);

  auto t = spmma(A.data().get(), B.data().get(), C.data().get(), m, n, k, batch_size);
  std::cout << "Pruning Time (ms): " << t[0] << std::endl;
  std::cout << "Compression Time (ms): " << t[1] << std::endl;
  std::cout << "SpMMA Time (ms): " << t[2] << std::endl;
  return EXIT_SUCCESS;
}
