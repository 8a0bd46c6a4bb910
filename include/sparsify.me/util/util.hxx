// util.hxx -- host helpers with the reference's names (include/sparsify.me/util/util.hxx:20-61):
// get_random, ceil_div, mat_sz, read_shapes.  Like the reference it pulls in timer.hxx and
// launch.hxx at its end (reference :66-67).
#pragma once
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

namespace sparsifyme {
namespace util {

// One uniform draw from [begin, end).  The reference builds and seeds a fresh std::mt19937 from
// std::random_device on EVERY call (util.hxx:21-26), which makes inputs irreproducible and costs
// microseconds per element; this build keeps one engine per thread, seeded once the same way.
// Call seed_random() for reproducible inputs.
inline std::mt19937& random_engine() {
  thread_local std::mt19937 gen{std::random_device{}()};
  return gen;
}
inline void seed_random(unsigned seed) { random_engine().seed(seed); }

template <typename type_t = float>
type_t get_random(type_t begin = 0.0f, type_t end = 1.0f) {
  std::uniform_real_distribution<> dis(static_cast<double>(begin), static_cast<double>(end));
  return static_cast<type_t>(dis(random_engine()));
}

template <typename type_t>
type_t ceil_div(type_t x, type_t y) {
  return (x + y - 1) / y;
}

// m, n, k, b
typedef std::tuple<int, int, int, int> mat_sz;

// Reads a shape table (header line, then `m,n,k,b` per line; CRLF tolerated).  Throws
// `const char*` when the file cannot be opened, as the reference does (util.hxx:41).
inline std::vector<mat_sz> read_shapes(std::string filename) {
  std::ifstream in(filename);
  if (!in.is_open()) throw "Unable to open shape CSV file.";
  std::vector<mat_sz> shapes;
  std::string line;
  std::getline(in, line);  // header
  while (std::getline(in, line)) {
    if (line.empty() || line == "\r") continue;
    std::istringstream fields(line);
    std::string f;
    int v[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4 && std::getline(fields, f, ','); ++i) v[i] = std::stoi(f);
    shapes.push_back(std::make_tuple(v[0], v[1], v[2], v[3]));
  }
  return shapes;
}

}  // namespace util
}  // namespace sparsifyme

#include <sparsify.me/util/timer.hxx>
#include <sparsify.me/util/launch.hxx>
#include <sparsify.me/util/alias.hxx>  // namespace sparsify = sparsifyme, only with -DSPARSIFYME_NAMESPACE_ALIAS
