// coo_fast.h -- pieces shared by the two matrix-core forms of the COO x dense-batch product (sm_spmm_coo_f32_fast): the dense-MFMA
// pipeline in spmm.hip and the sparse-matrix-instruction kernel in spmm_coo_smfmac.hip.
#pragma once
#include "sm_common.h"

namespace sm {

// Header of the fast form's workspace (round 4: power-of-two operand scales, so that the 2^-11 bound does not depend on the
// operands' magnitude, and a range flag, so that a caller can tell when it did not hold).  All written on the device.
struct CooFastHdr {
  int flag;            // != 0: an element left the fp16 range under the call's scales -> C was NOT written (the GEMM returns at once)
  unsigned max_b;      // bit pattern of max |b| over a strided sample of the dense operand (the scale's ESTIMATE)
  unsigned max_a;      // bit pattern of max |a| over all values of A (exact)
  float inv_scale[2];                  // 2^-y, 2^-x (the inverse scales of A and of the dense operand), written by the split kernel,
                                       // applied to the fp32 sums one after the other (each a normal float)
};
constexpr size_t COO_FAST_HDR_BYTES = 256;

// scale = 2^(target - floor(log2(max))), kept a finite normal float; max = 0 / inf / NaN -> 1 (the conversion pass then flags inf / NaN)
__device__ __forceinline__ int coo_fast_scale_exp(unsigned maxbits, int target) {
  const int e = (int)(maxbits >> 23) - 127;
  if (maxbits == 0u || maxbits >= 0x7f800000u) return 0;
  int x = target - (e < -126 ? -126 : e);
  x = x > 126 ? 126 : (x < -126 ? -126 : x);
  return x;
}
// sampled max |b| -> [2^12, 2^13): elements up to 8 x the sample's maximum still convert; max |a| -> [2^13, 2^14): room for
// duplicates that add up.  Every kernel derives the scales from the two maxima itself (no separate launch).
__device__ __forceinline__ float coo_fast_pow2(int x) { return __builtin_bit_cast(float, (unsigned)(x + 127) << 23); }
// |x * scale| must not exceed fp16's largest finite value (and x must be finite): then fp16(x * scale) has relative error
// <= 2^-11 in the normal range and ABSOLUTE error <= 2^-25 below it (|x * scale| < 2^-14: more than 2^26 below the operand's
// largest element) -- the bound stated in include/sparsifyme.h.  Underflow is therefore not flagged: with ~10^8 elements of
// ordinary data a few always fall that far below the maximum, and what they lose is 2^-37 of the maximum each.
__device__ __forceinline__ bool coo_fast_out_of_range(float xs) { return !(__builtin_fabsf(xs) <= 65504.0f); }

// spmm.hip: max |a| (exact) and a sampled max |b| into the header (which the caller has zeroed on the stream)
void coo_fast_scan(const float* vals, size_t nnz, const float* B, size_t nB, CooFastHdr* hdr, hipStream_t st);
// spmm_coo_smfmac.hip: the product on v_smfmac_f32_16x16x64_f16 (beta == 0; rows % 4 == 0); workspace bytes it needs (0: shape not taken)
size_t coo_smfmac_workspace(size_t m, size_t k, size_t nv);
bool coo_smfmac_takes(size_t m, size_t k, size_t nnz, size_t nv, const float* B, const float* C, float beta);
int coo_smfmac_product(size_t m, size_t k, size_t nnz, size_t nv, const int* rows, const int* cols, const float* vals, const float* B, float* C, float alpha,
                       void* workspace, hipStream_t st);

}  // namespace sm
