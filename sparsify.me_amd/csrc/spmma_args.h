// spmma_args.h -- argument block shared by the 2:4 matmul kernels (spmma_f16.hip, spmma_f16_pc.hip).
#pragma once
#include "mma_tile.h"

namespace sm {

// A launch serves up to SPMMA_MAXG same-shape problems (sm_spmma_*_grouped: the 3-6 instances of one layer shape in a network as ONE
// grid, so that a few-tile shape's last partial round is filled by the next instance): the operand pointers are tables indexed by the
// problem a workgroup belongs to, block -> (problem, grid batch, tile).  A single call is a group of one.
constexpr int SPMMA_MAXG = 8;
struct SpmmaArgs {
  const char* vals[SPMMA_MAXG];   // stage-major [kc/64][Mtot][32] halves (64 B per row per plane)
  const char* meta[SPMMA_MAXG];   // stage-major [kc/64][Mtot][8 B]
  size_t Mtot;        // rows of the whole blob (m * batch)
  const half_t* B[SPMMA_MAXG];
  half_t* C[SPMMA_MAXG];
  int ngroup;
  size_t sB, sC;      // batch strides (elements); rows of batch b are [b*m, (b+1)*m)
  int m;              // rows per batch
  int Mrows;          // rows this launch treats as one matrix (m, or m*batch when stacked)
  int N, K, kc;
  int batch;          // grid batches (1 when stacked)
  int tiles_m, tiles_n;
  float alpha, beta;
#ifdef SM_STAMP
  unsigned long long* dbg;  // diagnostic build only: per-wave cycle sums (never in the product library)
#endif
};


// The dense twin (spmma_f16_fused.hip): C = alpha * A * B + beta * C, row-major, dense, through the pipelines of the fused 2:4
// kernels (direct / big / span) with dense MFMA in place of selection + SMFMAC -- so that the dense GEMM the 2:4 path is measured
// against is not held back by a weaker pipeline on the shapes where those pipelines are the better ones (ragged k: the span form;
// n > 128: 256 x 256 tiles).  SM_STATUS_NOT_SUPPORTED: not a shape it serves; the caller runs gemm_f16.hip's own kernels.
struct DenseTwinCall {
  const half_t* A; const half_t* B; half_t* C;
  const half_t* const* Ap; const half_t* const* Bp; half_t* const* Cp;  // optional device pointer arrays (batch entries)
  size_t sA, sB, sC;
  int M, N, K, lda, batch;
  float alpha, beta;
  bool bf;
  int mode;  // 1: where the rule says the twin wins; 2: wherever it is supported (tuning)
  void* workspace;         // sm_gemm_*_ws: the stream-K form may run (sm_spmma_fused_workspace_size bytes, zero flag page); else null
  size_t workspace_bytes;
};
int gemm_dense_twin(const DenseTwinCall& c, hipStream_t st);

// spmma_f16_thin.hip: the fused 2:4 product for n < 8, k <= 64 (depthwise convolutions as im2col products) on the vector ALUs;
// SM_STATUS_NOT_SUPPORTED for anything else
int spmma_fused_thin(bool bf, int ngroup, const void* const* A, const void* const* B, void* const* C, size_t rows, size_t n, size_t k,
                     float alpha, float beta, hipStream_t st);

// spmma_f16_fused.hip: does sm_spmma_fused_{f16,bf16} take this single problem with one of its EXACT forms (bit-identical to
// sm_compress24 + sm_spmma; the thin form is excluded)?  Asked by the prune-in-place + multiply entry points before they touch A.
bool spmma_fused16_takes_exact(const void* A, const void* B, const void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB,
                               size_t strideC);

#ifdef SM_STAMP
__device__ __forceinline__ unsigned long long sm_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define SM_T(...) __VA_ARGS__
#else
#define SM_T(...)
#endif


}  // namespace sm
