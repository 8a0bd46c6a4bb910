// spmma_ops m n k b -- sparsifyme::spmma with transposed operands (reference include/sparsify.me/spmma.hxx:30-31,
// 67-69: transpose_a / transpose_b reach the matmul descriptor; no reference driver passes them).  Builds op(A) and
// op(B) once, stores them transposed, runs spmma() for the four (transpose_a, transpose_b) pairs and compares C and
// the in-place pruned A with the (N, N) call bit for bit.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <sparsify.me/containers/vector.hxx>
#include <sparsify.me/spmma.hxx>
#include <sparsify.me/util/gen.hxx>
#include <sparsify.me/util/util.hxx>

int main(int argc, char** argv) {
  using namespace sparsifyme;
  using type_t = _Float16;
  if (argc != 5) {
    std::cout << "Invalid # of arguments. Usage: ./spmma_ops m n k b" << std::endl;
    return EXIT_FAILURE;
  }
  const std::size_t m = std::stoi(argv[1]), n = std::stoi(argv[2]), k = std::stoi(argv[3]), b = std::stoi(argv[4]);
  device_vector<type_t> A(m * k * b), B(k * n * b), At(m * k * b), Bt(k * n * b);
  util::random::uniform_distribution(A, -1.0f, 1.0f, 11);
  util::random::uniform_distribution(B, -1.0f, 1.0f, 12);
  (void)sm_transpose(A.data().get(), At.data().get(), m, k, k, m, sizeof(type_t), b, m * k, m * k, nullptr);  // stored k x m
  (void)sm_transpose(B.data().get(), Bt.data().get(), k, n, n, k, sizeof(type_t), b, k * n, k * n, nullptr);  // stored n x k
  (void)hipDeviceSynchronize();
  const host_vector<type_t> hA0 = A.to_host(), hAt0 = At.to_host();

  std::vector<std::uint16_t> c_ref, a_ref;
  bool all = true;
  for (int ta = 0; ta < 2; ++ta)
    for (int tb = 0; tb < 2; ++tb) {
      device_vector<type_t> a = ta ? hAt0 : hA0;  // fresh unpruned operand, in its stored orientation
      device_vector<type_t> C(m * n * b);
      const auto t = spmma(a.data().get(), (tb ? Bt : B).data().get(), C.data().get(), m, n, k, b,
                           ta ? operation_t::T : operation_t::N, tb ? operation_t::T : operation_t::N);
      host_vector<type_t> hC = C.to_host();
      std::vector<std::uint16_t> c(m * n * b), ap(m * k * b);
      std::memcpy(c.data(), hC.data(), c.size() * 2);
      if (ta) {  // bring the in-place pruned stored A back to the N form for the comparison
        device_vector<type_t> back(m * k * b);
        (void)sm_transpose(a.data().get(), back.data().get(), k, m, m, k, sizeof(type_t), b, m * k, m * k, nullptr);
        host_vector<type_t> hb = back.to_host();
        std::memcpy(ap.data(), hb.data(), ap.size() * 2);
      } else {
        host_vector<type_t> ha = a.to_host();
        std::memcpy(ap.data(), ha.data(), ap.size() * 2);
      }
      bool ok = true;
      if (!ta && !tb) { c_ref = c; a_ref = ap; }
      else ok = c == c_ref && ap == a_ref;
      all = all && ok;
      std::cout << "transpose_a=" << (ta ? "T" : "N") << " transpose_b=" << (tb ? "T" : "N") << " ms " << t[0] << " " << t[1] << " "
                << t[2] << " matches N,N: " << (ok ? "yes" : "NO") << std::endl;
    }
  return all ? EXIT_SUCCESS : EXIT_FAILURE;
}
