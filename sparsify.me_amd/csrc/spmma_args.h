// spmma_args.h -- argument block shared by the 2:4 matmul kernels (spmma_f16.hip, spmma_f16_pc.hip).
#pragma once
#include "mma_tile.h"

namespace sm {

struct SpmmaArgs {
  const char* vals;   // stage-major [kc/64][Mtot][32] halves (64 B per row per plane)
  const char* meta;   // stage-major [kc/64][Mtot][8 B]
  size_t Mtot;        // rows of the whole blob (m * batch)
  const half_t* B;
  half_t* C;
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
