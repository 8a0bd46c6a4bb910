// spmma_f16_thin.hip -- the fused 2:4 product for THIN problems (round 5): n < 8 output columns and k <= 64, i.e. the depthwise
// convolutions of the reference's model zoo (datasets/get_shapes.py:87-98: MobileNetV2 / V3 -- as im2col products m = out_h * out_w,
// n = 1, k = 9 or 25, b = batch x channels up to 30 720).  C = alpha * prune24_strip(A) * B + beta * C with A one tall contiguous
// matrix (lda == k, the batches stacked, B shared).
//
// Why its own kernel: the matrix-core kernels pad such a problem to a 128 x 64 x 64 tile (0.2 % of the multiply-adds useful), and the
// staged pair first writes a blob 3.5 x the size of A (a 64-byte value plane + 8 bytes of metadata per 18-byte row).  The product is
// pure streaming -- 18 to 50 bytes of A per output element -- so it runs on the vector ALUs: a workgroup brings ROWS consecutive rows
// (one contiguous span of ROWS * k * 2 bytes, 16-byte loads whatever the row pitch) to LDS, then every thread takes rows of its own:
// k two-byte LDS reads, the frozen STRIP rule per strip of four (select24.h: strip_keepmask; a ragged last strip is completed with
// virtual zeros, oracle: strip_select), fp32 multiply-adds against B (broadcast reads of the k x n operand in LDS) in ascending k,
// one rounding to the 16-bit output.  Result: inside the tight bound of the fp64 product (tests: the oracle); NOT bit-identical to
// sm_compress24 + sm_spmma, whose matrix instruction adds the same products in another order (stated in include/sparsifyme.h).
#include "select24.h"
#include "spmma_args.h"

namespace sm {

struct ThinArgs {
  const half_t* A[8];
  const half_t* B[8];
  half_t* C[8];
  unsigned long long rows;   // rows of the tall matrix (m * batch)
  int N, K;
  float alpha, beta;
};

template <bool BF>
__device__ __forceinline__ float thin_f32(unsigned short bits) {
  if constexpr (BF) return __builtin_bit_cast(float, (uint32_t)bits << 16);
  else return (float)__builtin_bit_cast(half_t, bits);
}

constexpr int THIN_ROWS = 1024;  // rows per workgroup (four per thread)

template <bool BF>
__global__ __launch_bounds__(256) void spmma_f16_thin_kernel(const ThinArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned tid = threadIdx.x;
  const unsigned g = blockIdx.y;
  const char* A = reinterpret_cast<const char*>(p.A[0]);
  const half_t* B = p.B[0];
  half_t* C = p.C[0];
#pragma unroll
  for (unsigned i = 1; i < 8; ++i)
    if (g == i) { A = reinterpret_cast<const char*>(p.A[i]); B = p.B[i]; C = p.C[i]; }
  const unsigned K = (unsigned)p.K, N = (unsigned)p.N, rowbytes = K * 2u;
  const unsigned long long row0 = (unsigned long long)blockIdx.x * THIN_ROWS;
  const unsigned nrows = p.rows - row0 < (unsigned long long)THIN_ROWS ? (unsigned)(p.rows - row0) : (unsigned)THIN_ROWS;
  // ---- the span: bytes [row0 * rowbytes, (row0 + nrows) * rowbytes) in 16-byte pieces (row0 * rowbytes is a multiple of 2048);
  //      the last piece of the operand is re-read by the lanes past it (a_bytes % 16 == 0: launcher)
  const unsigned long long a_bytes = p.rows * rowbytes, s0 = row0 * rowbytes;
  const unsigned len = nrows * rowbytes, np = (len + 15u) / 16u;
  for (unsigned pc = tid; pc < np; pc += 256u) {
    unsigned long long off = s0 + (unsigned long long)pc * 16u;
    off = off < a_bytes - 16u ? off : a_bytes - 16u;
    const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(A + off));
    // (a clamped piece lands at its own slot: the rows that read it are past the operand's end and never stored)
    *reinterpret_cast<u4*>(smem + (size_t)pc * 16u) = v;
  }
  // ---- B (k x n, row-major) behind the span, as fp32
  float* Bs = reinterpret_cast<float*>(smem + (size_t)THIN_ROWS * rowbytes + 16);
  for (unsigned i = tid; i < K * N; i += 256u) Bs[i] = thin_f32<BF>(__builtin_bit_cast(unsigned short, B[i]));
  __syncthreads();
  const unsigned nstrips = (K + 3u) / 4u;
  for (unsigned r = tid; r < nrows; r += 256u) {
    const char* src = smem + (size_t)r * rowbytes;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (unsigned s = 0; s < nstrips; ++s) {
      unsigned short x[4];
#pragma unroll
      for (unsigned e = 0; e < 4; ++e) {
        const unsigned kk = 4u * s + e;
        x[e] = kk < K ? *reinterpret_cast<const unsigned short*>(src + 2u * kk) : (unsigned short)0;  // virtual zeros complete a ragged strip
      }
      const unsigned keep = strip_keepmask(key_of(x[0]), key_of(x[1]), key_of(x[2]), key_of(x[3]));
#pragma unroll
      for (unsigned e = 0; e < 4; ++e) {
        const unsigned kk = 4u * s + e;
        if (((keep >> e) & 1u) && kk < K) {
          const float a = thin_f32<BF>(x[e]);
#pragma unroll
          for (unsigned j = 0; j < 7; ++j)   // (constant indices: the sums stay in registers)
            if (j < N) acc[j] = __builtin_fmaf(a, Bs[kk * N + j], acc[j]);
        }
      }
    }
    half_t* dst = C + (row0 + r) * N;
#pragma unroll
    for (unsigned j = 0; j < 7; ++j)
      if (j < N) {
        float v = p.alpha * acc[j];
        if (p.beta != 0.0f) v += p.beta * to_f32<BF>(dst[j]);
        dst[j] = to_elt<BF>(v);
      }
  }
}

// n < 8, k <= 64, one tall contiguous A per problem (rows = m * batch, lda == k), shared B; up to 8 same-shape problems per launch.
int spmma_fused_thin(bool bf, int ngroup, const void* const* A, const void* const* B, void* const* C, size_t rows, size_t n, size_t k,
                     float alpha, float beta, hipStream_t st) {
  if (n == 0 || n >= 8 || k == 0 || k > 64 || ngroup < 1 || ngroup > 8 || (rows * k * 2) % 16 != 0 || rows * k * 2 < 16) return SM_STATUS_NOT_SUPPORTED;
  for (int g = 0; g < ngroup; ++g)
    if (!aligned16(A[g]) || (reinterpret_cast<uintptr_t>(B[g]) & 1u) || (reinterpret_cast<uintptr_t>(C[g]) & 1u)) return SM_STATUS_NOT_SUPPORTED;
  ThinArgs a = {};
  for (int g = 0; g < 8; ++g) {
    const int s_ = g < ngroup ? g : 0;
    a.A[g] = (const half_t*)A[s_]; a.B[g] = (const half_t*)B[s_]; a.C[g] = (half_t*)C[s_];
  }
  a.rows = rows; a.N = (int)n; a.K = (int)k; a.alpha = alpha; a.beta = beta;
  const size_t nblk = (rows + THIN_ROWS - 1) / THIN_ROWS;
  if (nblk > 0x7fffffffu) {
    set_error("sm_spmma_fused_{f16,bf16}: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  const size_t lds = (size_t)THIN_ROWS * k * 2 + 16 + k * n * 4 + 16;   // <= 128 KiB + 1.8 KiB
  static LdsOptIn optin_h, optin_b;
  if (bf) {
    if (const int rc = ensure_dyn_lds(optin_b, reinterpret_cast<const void*>(&spmma_f16_thin_kernel<true>), 160 * 1024, "spmma_f16_thin_kernel")) return rc;
    spmma_f16_thin_kernel<true><<<dim3((unsigned)nblk, (unsigned)ngroup), dim3(256), lds, st>>>(a);
  } else {
    if (const int rc = ensure_dyn_lds(optin_h, reinterpret_cast<const void*>(&spmma_f16_thin_kernel<false>), 160 * 1024, "spmma_f16_thin_kernel")) return rc;
    spmma_f16_thin_kernel<false><<<dim3((unsigned)nblk, (unsigned)ngroup), dim3(256), lds, st>>>(a);
  }
  return check_launch("spmma_f16_thin_kernel");
}

}  // namespace sm
