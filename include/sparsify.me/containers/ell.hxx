// ell.hxx -- Blocked-ELL sparse matrix container.
// Same fields and meaning as the reference's include/sparsify.me/containers/ell.hxx:23-33:
//   rows x cols matrix cut into block_size x block_size blocks; every block row stores
//   blocked_cols = ell_cols / block_size blocks; column_indices[blocked_rows][blocked_cols] holds
//   each stored block's block-column; values[rows][ell_cols] row-major holds the blocks' elements.
// Assignment between memory spaces copies the arrays across (reference :38-50).
#pragma once
#include <cstddef>
#include <iostream>

#include <sparsify.me/containers/memory.hxx>
#include <sparsify.me/containers/vector.hxx>

namespace sparsifyme {
template <typename type_t = float, memory_space_t space = memory_space_t::device>
struct ell_t {
  std::size_t rows = 0, cols = 0, block_size = 0;
  std::size_t ell_cols = 0;
  std::size_t blocked_rows = 0;  // rows / block_size
  std::size_t blocked_cols = 0;  // ell_cols / block_size
  std::size_t num_blocks = 0;    // blocked_rows * blocked_cols

  vector_t<std::size_t, space> column_indices;  // [blocked_rows x blocked_cols]
  vector_t<type_t, space> values;               // [rows x ell_cols]

  ell_t() {}
  ~ell_t() {}

  template <memory_space_t in_space>
  ell_t<type_t, space>& operator=(const ell_t<type_t, in_space>& rhs) {
    rows = rhs.rows;
    cols = rhs.cols;
    block_size = rhs.block_size;
    ell_cols = rhs.ell_cols;
    blocked_rows = rhs.blocked_rows;
    blocked_cols = rhs.blocked_cols;
    num_blocks = rhs.num_blocks;
    assign(column_indices, rhs.column_indices);
    assign(values, rhs.values);
    return *this;
  }

  void print() {
    std::cout << "A-Matrix" << std::endl;
    std::cout << "\t(rows, cols) = " << rows << ", " << cols << std::endl;
    std::cout << "\tELL columns = " << ell_cols << std::endl;
    std::cout << "\tBlock Size = " << block_size << std::endl;
    std::cout << "\tNumber of Blocks = " << num_blocks << std::endl;
    host_vector<std::size_t> ci;
    assign(ci, column_indices);
    std::cout << "\tColumn Idx = ";
    for (auto c : ci) std::cout << c << " ";
    std::cout << std::endl;
    host_vector<type_t> v;
    assign(v, values);
    std::cout << "\tValues = ";
    for (auto& x : v) std::cout << static_cast<float>(x) << " ";
    std::cout << std::endl;
  }
};
}  // namespace sparsifyme
