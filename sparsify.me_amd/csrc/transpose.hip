// transpose.hip -- out-of-place 2-D transpose of row-major matrices (2-, 4- or 8-byte elements, strided batch).
// Serves the transposed operands of sparsifyme::spmma (reference include/sparsify.me/spmma.hxx:30-31,67-69 hands
// transpose_a / transpose_b to the vendor's matmul descriptor): the header brings a transposed operand to the N form
// the 2:4 kernels consume, and writes the pruned A back in its stored orientation.  HBM-bound: 64 x 64 element tiles
// through LDS.  transpose_vec_kernel (leading dimensions, batch strides and bases multiples of 16 bytes): 16-byte global
// loads and stores on both sides, the tile written to LDS as 16-byte row pieces and read back element-wise along columns;
// transpose_kernel (anything else): element-wise accesses throughout.
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

// 16-byte global accesses on both sides (V = 16 / sizeof(T) elements per access); edges are handled per chunk, with an
// element-wise tail where a chunk straddles the matrix edge.
template <typename T>
__global__ __launch_bounds__(256) void transpose_vec_kernel(const T* __restrict__ in, T* __restrict__ out, size_t rows, size_t cols,
                                                            size_t ld_in, size_t ld_out, size_t stride_in, size_t stride_out) {
  constexpr unsigned V = 16 / sizeof(T), CPR = 64 / V;  // chunks per 64-element row
  constexpr unsigned PITCH = 64 + V;                    // elements; rows stay 16-byte aligned, columns spread over banks
  __shared__ __attribute__((aligned(16))) T tile[64 * PITCH];
  const size_t r0 = (size_t)blockIdx.y * 64, c0 = (size_t)blockIdx.x * 64;
  in += (size_t)blockIdx.z * stride_in;
  out += (size_t)blockIdx.z * stride_out;
  for (unsigned i = threadIdx.x; i < 64 * CPR; i += 256) {
    const unsigned r = i / CPR, c = (i % CPR) * V;
    u4 v = {0u, 0u, 0u, 0u};
    if (r0 + r < rows) {
      const T* src = in + (r0 + r) * ld_in + c0 + c;
      if (c0 + c + V <= cols) v = *reinterpret_cast<const u4*>(src);
      else {
        T e[V];
        for (unsigned t = 0; t < V; ++t) e[t] = c0 + c + t < cols ? src[t] : T(0);
        v = *reinterpret_cast<const u4*>(e);
      }
    }
    *reinterpret_cast<u4*>(&tile[r * PITCH + c]) = v;
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < 64 * CPR; i += 256) {
    const unsigned c = i / CPR, r = (i % CPR) * V;  // output row c0 + c, elements r0 + r .. + V - 1 along it
    if (c0 + c >= cols || r0 + r >= rows) continue;
    T e[V];
#pragma unroll
    for (unsigned t = 0; t < V; ++t) e[t] = tile[(r + t) * PITCH + c];
    T* dst = out + (c0 + c) * ld_out + r0 + r;
    if (r0 + r + V <= rows) *reinterpret_cast<u4*>(dst) = *reinterpret_cast<const u4*>(e);
    else
      for (unsigned t = 0; t < V && r0 + r + t < rows; ++t) dst[t] = e[t];
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
  const size_t v = 16 / elt_bytes;
  if (aligned16(in) && aligned16(out) && ld_in % v == 0 && ld_out % v == 0 && (batch == 1 || (stride_in % v == 0 && stride_out % v == 0))) {
    if (elt_bytes == 2)
      transpose_vec_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)in, (uint16_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
    else if (elt_bytes == 4)
      transpose_vec_kernel<uint32_t><<<grid, 256, 0, st>>>((const uint32_t*)in, (uint32_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
    else
      transpose_vec_kernel<uint64_t><<<grid, 256, 0, st>>>((const uint64_t*)in, (uint64_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
    return check_launch("transpose_vec_kernel");
  }
  if (elt_bytes == 2)
    transpose_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)in, (uint16_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
  else if (elt_bytes == 4)
    transpose_kernel<uint32_t><<<grid, 256, 0, st>>>((const uint32_t*)in, (uint32_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
  else
    transpose_kernel<uint64_t><<<grid, 256, 0, st>>>((const uint64_t*)in, (uint64_t*)out, rows, cols, ld_in, ld_out, stride_in, stride_out);
  return check_launch("transpose_kernel");
}
