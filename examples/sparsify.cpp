// sparsify m n -- times sparsifyme::sparsify<2,2> on an m x n matrix (the CLI and the one-line
// output of the reference's examples/sparsify.cu:19-54: elapsed milliseconds).
#include <cstdlib>
#include <iostream>
#include <string>

#include <sparsify.me/containers/vector.hxx>
#include <sparsify.me/sparsify.hxx>
#include <sparsify.me/util/util.hxx>

#ifndef SM_TYPE
#define SM_TYPE float
#endif

int main(int argc, char** argv) {
  using namespace sparsifyme;
  using type_t = SM_TYPE;
  if (argc != 3) {
    std::cout << "Invalid # of arguments. Usage: ./sparsify m n" << std::endl;
    return EXIT_FAILURE;
  }
  std::size_t m = std::stoi(argv[1]), n = std::stoi(argv[2]);

  host_vector<type_t> h_weights(m * n);
  for (auto& w : h_weights) w = static_cast<type_t>(util::get_random<float>());
  device_vector<type_t> weights = h_weights;
  device_vector<std::size_t> mask(m * n);

  util::timer_t timer;
  timer.begin();
  sparsify<2, 2>(weights.data().get(), mask.data().get(), m, n);
  timer.end();
  std::cout << timer.milliseconds() << std::endl;
  return EXIT_SUCCESS;
}
