// spmma m n k b -- prune to 2:4, compress, multiply; prints the three stage times with the labels
// of the reference's examples/spmma.cu:64-66.  The reference refuses every GPU that is not compute
// capability 8.0 (:35-41); this driver refuses every GPU that is not gfx950.  Element type: fp16,
// the type the reference's spmma actually declares to its back end (spmma.hxx:40); build with
// -DSM_TYPE=float for the fp32 kernels (bin/spmma_f32; there a fifth argument 3 / 2 sets spmma_options().f32_planes: the multiply
// on the sparse matrix instruction through exact bfloat16 splits instead of dense fp32 matrix work, include/sparsifyme.h).
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
  if (argc != 5 && argc != 6) {
    std::cout << "Invalid # of arguments. Usage: ./spmma m n k b" << (sizeof(type_t) == 4 ? " [f32_planes: 0 | 2 | 3]" : " [fewest]") << std::endl;
    return EXIT_FAILURE;
  }
  if (argc == 6 && sizeof(type_t) != 4) {
    // optional fifth argument of the 16-bit drivers: "fewest" = spmma_options().fewest_passes (the whole sequence through
    // sm_prune24_spmma_*: one measured time, printed first; the other two lines are 0).  Default: three measured stage times.
    if (std::string(argv[5]) != "fewest") {
      std::cout << "Invalid # of arguments. Usage: ./spmma m n k b [fewest]" << std::endl;
      return EXIT_FAILURE;
    }
    spmma_options().fewest_passes = true;
  } else if (argc == 6) {
    const int planes = std::stoi(argv[5]);
    if (planes != 0 && planes != 2 && planes != 3) {
      std::cout << "f32_planes must be 0, 2 or 3" << std::endl;
      return EXIT_FAILURE;
    }
    spmma_options().f32_planes = planes;
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
