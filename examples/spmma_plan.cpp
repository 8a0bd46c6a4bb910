// spmma_plan m n k b [reps] -- the cached-plan form of spmma (sparsifyme::spmma_plan_t, an extension of this build;
// SURVEY.md 8(f) rank 1): compress the 2:4 operand once, multiply `reps` times against changing B.  Prints the
// one-off compression time, the per-call multiply time, and checks on the device data that the plan's first
// product is bit-identical to what sparsifyme::spmma() -- which re-creates the blob per call, as the reference does
// (include/sparsify.me/spmma.hxx:97-113) -- returns for the same operands.
#include <cstdlib>
#include <cstring>
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
    std::cout << "Invalid # of arguments. Usage: ./spmma_plan m n k b [reps]" << std::endl;
    return EXIT_FAILURE;
  }
  if (sm_device_check() != SM_STATUS_SUCCESS) {
    std::cerr << "\nlibsparsifyme is supported only on gfx950 (MI355X) devices: " << sm_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  const std::size_t m = std::stoi(argv[1]), n = std::stoi(argv[2]), k = std::stoi(argv[3]), b = std::stoi(argv[4]);
  const int reps = argc == 6 ? std::stoi(argv[5]) : 10;

  host_vector<type_t> h_A(m * k * b), h_B(k * n * b);
  for (auto& a : h_A) a = static_cast<type_t>(util::get_random<float>());
  for (auto& x : h_B) x = static_cast<type_t>(util::get_random<float>());
  device_vector<type_t> A = h_A, A2 = h_A, B = h_B, C(m * n * b), C2(m * n * b);

  // the per-call path: prune in place (TILE), check, compress into a temporary, multiply
  (void)spmma(A2.data().get(), B.data().get(), C2.data().get(), m, n, k, b);

  spmma_plan_t<type_t> plan(m, k, b);
  util::timer_t t;
  t.begin();
  int rc = plan.compress(A.data().get(), /*prune_in_place=*/true);
  const float compress_ms = t.end();
  t.begin();
  for (int r = 0; r < reps && rc == SM_STATUS_SUCCESS; ++r) rc = plan.multiply(B.data().get(), C.data().get(), n);
  const float mul_ms = t.end() / (reps > 0 ? reps : 1);
  if (rc != SM_STATUS_SUCCESS) {
    std::cerr << "spmma_plan: " << sm_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  (void)hipDeviceSynchronize();
  const host_vector<type_t> h_C = C.to_host(), h_C2 = C2.to_host();
  const bool same = std::memcmp(h_C.data(), h_C2.data(), h_C.size() * sizeof(type_t)) == 0;
  std::cout << "Compressed bytes: " << plan.compressed_bytes() << std::endl;
  std::cout << "Compression Time (ms): " << compress_ms << std::endl;
  std::cout << "SpMMA Time (ms): " << mul_ms << std::endl;
  std::cout << "Matches spmma(): " << (same ? "yes" : "NO") << std::endl;
  return same ? EXIT_SUCCESS : EXIT_FAILURE;
}
