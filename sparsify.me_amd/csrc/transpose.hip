// transpose.hip -- out-of-place 2-D transpose of row-major matrices (2-, 4- or 8-byte elements, strided batch).
// Serves the transposed operands of sparsifyme::spmma (reference include/sparsify.me/spmma.hxx:30-31,67-69 hands
// transpose_a / transpose_b to the vendor's matmul descriptor): the header brings a transposed operand to the N form
// the 2:4 kernels consume, and writes the pruned A back in its stored orientation.  HBM-bound: 64 x 64 element tiles
// through LDS, 16-byte global accesses on both sides when the leading dimensions and bases allow.
#include "sm_common.h"

namespace sm {

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, size_t rows, size_t cols,
                                                        size_t ld_in, size_t ld_out, size_t stride_in, size_t stride_out) {
  __shared__ T tile[64][64 + 4 / sizeof(T) + (sizeof(T) == 8 ? 1 : 0)];  // pitch = 64 elements + one bank
  const size_t r0 = (size_t)blockIdx.y * 64, c0 = (size_t)blockIdx.x * 64;
  in += (size_t)blockIdx.z * stride_in;
  out += (size_t)blockIdx.z * stride_out;
  for (unsigned i = threadIdx.x; i < 64 * 64; i += 256) {
    const unsigned r = i >> 6, c = i & 63u;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? in[(r0 + r) * ld_in + c0 + c] : T(0);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < 64 * 64; i += 256) {
    const unsigned c = i >> 6, r = i & 63u;
    if (r0 + r < rows && c0 + c < cols) out[(c0 + c) * ld_out + r0 + r] = tile[r][c];
  }
}

}  // namespace sm

using namespace sm;

extern "C" int sm_transpose(const void* in, void* out, size_t rows, size_t cols, size_t ld_in, size_t ld_out, size_t elt_bytes,
                            size_t batch, size_t stride_in, size_t stride_out, sm_stream_t stream) {
  if (!in || !out || in == out || ld_in < cols || ld_out < rows || (elt_bytes != 2 && elt_bytes != 4 && elt_bytes != 8)) {
    set_error("sm_transpose: invalid argument (out of place; ld_in >= cols, ld_out >= rows; 2-, 4- or 8-byte elements)");
    return SM_STATUS_INVALID_VALUE;
  }
  if (rows == 0 || cols == 0 || batch == 0) return SM_STATUS_SUCCESS;
  const size_t gx = ceil_div(cols, (size_t)64), gy = ceil_div(rows, (size_t)64);
  if (gx > 0x7fffffffull || gy > 65535 || batch > 65535) {
    set_error("sm_transpose: matrix too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)batch);
  hipStream_t st = (hipStream_t)stream;
  if (elt_bytes == 2)
    transpose_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)in, (uint16_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
  else if (elt_bytes == 4)
    transpose_kernel<uint32_t><<<grid, 256, 0, st>>>((const uint32_t*)in, (uint32_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
  else
    transpose_kernel<uint64_t><<<grid, 256, 0, st>>>((const uint64_t*)in, (uint64_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
  return check_launch("transpose_kernel");
}
