/*
 * sparsifyme.h -- C ABI of libsparsifyme.so, the MI355X (gfx950) back end of the sparsify.me
 * hot path: prune to 2:4 -> compress -> sparse x dense matmul, measured against the library's
 * own dense batched GEMM.
 *
 * This is the drop-in boundary.  The reference (owensgroup/sparsify.me) is a header-only CUDA C++
 * template API; each entry point below replaces the vendor call(s) named next to it, and the
 * templates in include/sparsify.me/ *.hxx (same paths, namespace and signatures as the reference)
 * forward to these symbols.  Everything is `extern "C"`, plain pointers and sizes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - functions only enqueue work on `stream` (no allocation, no synchronisation), so a caller
 *     may capture them into a hipGraph (one exception, stated at sm_spmm_bell_batched_f32); they
 *     return SM_STATUS_* (0 = success) and never throw;
 *   - fp16 data are IEEE binary16 bit patterns (`_Float16` / `__half` / uint16_t storage).
 *
 * 2:4 compressed blob (produced by sm_compress24_*, consumed by sm_spmma_* / sm_decompress24_*),
 * for `batch` row-major m x k matrices, M = batch*m rows:
 *   kc       = k rounded up to a multiple of 64
 *   both sections are STAGE-major: plane s covers dense k 64s..64s+63 of every row
 *   values   : [kc/64][M][32] elements at byte 0 (kept pair of strip q of row R at
 *              [q/16][R][2(q%16)], [q/16][R][2(q%16)+1])
 *   metadata : [kc/64][M][8] bytes at byte round_up(M*(kc/2)*elt, 256); strip q's nibble
 *              (p0 | p1 << 2, p0 < p1 kept positions) in bits 4*(q&1).. of byte [q/16][R][(q%16)/2]
 *   size     = sm_compress24_size()
 */
#ifndef SPARSIFYME_H_
#define SPARSIFYME_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM_STATUS_SUCCESS 0
#define SM_STATUS_INVALID_VALUE 1   /* bad pointer / size / flag                          */
#define SM_STATUS_NOT_SUPPORTED 2   /* valid request this build has no kernel for          */
#define SM_STATUS_LAUNCH_FAILED 3   /* HIP reported an error at launch                     */
#define SM_STATUS_NO_DEVICE 4       /* no gfx950 device visible                            */

#define SM_PRUNE_TILE 0   /* cusparseLtPruneAlg_t numbering used at spmma.hxx:86 */
#define SM_PRUNE_STRIP 1

#define SM_OP_N 0
#define SM_OP_T 1

typedef void* sm_stream_t; /* hipStream_t */

/* Library / device info (host). */
const char* sm_version(void);
/* 0 when a gfx950 device is usable by this process, else SM_STATUS_NO_DEVICE. */
int sm_device_check(void);
/* Message of the last failing call made by the calling thread ("" if none). */
const char* sm_last_error(void);

/* ---- (a1) positional sparsify: replaces the Thrust fill + transform of
 *      include/sparsify.me/sparsify.hxx:71-81 (sparsifyme::sparsify<BLK_M,BLK_N,type_t>).
 *      weights: m*n elements of elt_bytes (2, 4 or 8), zeroed in place; mask: m*n uint64
 *      (the reference's std::size_t mask), written in full. */
int sm_sparsify_positional(void* weights, uint64_t* mask, size_t m, size_t n, size_t elt_bytes,
                           size_t blk_m, size_t blk_n, float sparsity_factor, sm_stream_t stream);
int sm_sparsify_positional_f16(void* weights, uint64_t* mask, size_t m, size_t n,
                               float sparsity_factor, sm_stream_t stream);
int sm_sparsify_positional_f32(float* weights, uint64_t* mask, size_t m, size_t n,
                               float sparsity_factor, sm_stream_t stream);
int sm_sparsify_positional_f64(double* weights, uint64_t* mask, size_t m, size_t n,
                               float sparsity_factor, sm_stream_t stream);

/* ---- (a2) prune to 2:4: replaces cusparseLtSpMMAPrune (spmma.hxx:86-87).
 *      A_in / A_out row-major m x k, leading dimension ld elements; may alias (in place). */
int sm_prune24_f16(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, int alg,
                   sm_stream_t stream);
int sm_prune24_f32(const float* A_in, float* A_out, size_t m, size_t k, size_t ld, int alg,
                   sm_stream_t stream);

/* ---- (a2) prune check: replaces cusparseLtSpMMAPruneCheck (spmma.hxx:88).
 *      *d_valid (device int) = 0 iff every 1x4 strip has <= 2 non-zeros, else 1. */
int sm_prune24_check_f16(const void* A, size_t m, size_t k, size_t ld, int* d_valid,
                         sm_stream_t stream);
int sm_prune24_check_f32(const float* A, size_t m, size_t k, size_t ld, int* d_valid,
                         sm_stream_t stream);

/* ---- (a3) compress: replaces cusparseLtSpMMACompressedSize + cusparseLtSpMMACompress
 *      (spmma.hxx:100-103).  A: batch matrices, row-major m x k, ld, batch stride strideA
 *      elements.  Keeps the two largest |x| of every strip (STRIP rule), i.e. for an already
 *      pruned A exactly its non-zeros; so compress alone is also the fused prune+compress. */
int sm_compress24_size(size_t m, size_t k, size_t elt_bytes, size_t batch, size_t* bytes /*host*/);
int sm_compress24_f16(const void* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA,
                      void* blob, sm_stream_t stream);
int sm_compress24_f32(const float* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA,
                      void* blob, sm_stream_t stream);
/* Inverse (test/debug aid; no reference counterpart). */
int sm_decompress24_f16(const void* blob, size_t m, size_t k, size_t ld, size_t batch,
                        size_t strideA, void* A, sm_stream_t stream);
int sm_decompress24_f32(const void* blob, size_t m, size_t k, size_t ld, size_t batch,
                        size_t strideA, float* A, sm_stream_t stream);

/* ---- (a2 + a3 in one pass) prune -> check -> compress as sparsifyme::spmma() runs them (spmma.hxx:82-104:
 *      cusparseLtSpMMAPrune in place, cusparseLtSpMMAPruneCheck, cusparseLtSpMMACompress): reads A_in once and writes
 *      the pruned operand to A_out (may alias A_in: in place; NULL: not wanted), the compressed blob (NULL: not wanted)
 *      and *d_valid (NULL: not wanted; 0 iff every strip written holds <= 2 non-zeros).  `batch` row-major m x k
 *      matrices strideA elements apart; a 4 x 4 TILE never straddles two of them.  alg = SM_PRUNE_TILE or
 *      SM_PRUNE_STRIP.  Same bytes as sm_prune24_* followed by sm_prune24_check_* and sm_compress24_* (which is what
 *      runs for shapes the one-pass kernel does not take: k % 64 != 0 or unaligned rows). */
int sm_prune24_compress24_f16(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, size_t batch, size_t strideA,
                              void* blob, int* d_valid, int alg, sm_stream_t stream);
int sm_prune24_compress24_bf16(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, size_t batch, size_t strideA,
                               void* blob, int* d_valid, int alg, sm_stream_t stream);
/* fp32 form (the type the reference's own driver instantiates, examples/spmma.cu:24): 2:4 on fp32 as everywhere in this
 * build; same bytes as sm_prune24_f32 + sm_prune24_check_f32 + sm_compress24_f32. */
int sm_prune24_compress24_f32(const float* A_in, float* A_out, size_t m, size_t k, size_t ld, size_t batch, size_t strideA,
                              void* blob, int* d_valid, int alg, sm_stream_t stream);

/* ---- (a2 + a4 without a blob, rounds 4 + 6) the whole call sequence of sparsifyme::spmma() (spmma.hxx:82-113: prune TILE in place,
 *      check, compress, multiply) with NO compressed blob: reads A_in, writes the pruned operand to A_out (A_in itself: in place, what
 *      the reference does; or a second buffer), raises *d_valid (NULL: not wanted; 0 iff every strip written holds <= 2 non-zeros,
 *      derived from the stored values) and computes C_b = alpha * prune24(A_b) * B_b + beta * C_b.  A_out is bit-identical to
 *      sm_prune24_*(A_in, alg) (per batch matrix when m % 4 != 0), C to sm_spmma_*(sm_compress24_*(A_out)).
 *      ONE kernel (A read once, written once) for n <= 128, n % 8 == 0, k % 64 == 0, m % 4 == 0 and 16-byte aligned rows.  Since round 6
 *      every other shape the EXACT fused kernels take -- n > 128 (k % 64 == 0, n % 8 == 0, 16-byte aligned rows), ragged k with
 *      n <= 128 on one contiguous A with a shared B (the span form), m % 4 != 0 -- runs as TWO launches inside the call: the prune + flag
 *      pass over A (sm_prune24_compress24_* with a null blob) and sm_spmma_fused_* on the pruned operand (its STRIP selection of a 2:4
 *      strip is what sm_compress24 stores for it); HBM bytes 2 A + C + B while the pruned A stays in the 256 MiB Infinity Cache between
 *      the two.  SM_STATUS_NOT_SUPPORTED -- decided BEFORE A is touched -- otherwise (n % 8 != 0, n < 8, unaligned operands): then
 *      sm_prune24_compress24_* + sm_spmma_*.  Any other failing status of the second launch (a refused LDS opt-in, a grid limit) is
 *      returned after A_out has been written. */
int sm_prune24_spmma_f16(const void* A_in, void* A_out, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                         size_t batch, size_t strideA, size_t strideB, size_t strideC, int alg, int* d_valid, float alpha,
                         float beta, sm_stream_t stream);
int sm_prune24_spmma_bf16(const void* A_in, void* A_out, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                          size_t batch, size_t strideA, size_t strideB, size_t strideC, int alg, int* d_valid, float alpha,
                          float beta, sm_stream_t stream);

/* ---- (a4) 2:4 sparse x dense matmul: replaces cusparseLtMatmul (spmma.hxx:112-113).
 *      C_b = alpha * A_b * B_b + beta * C_b, row-major, ld(B) = ld(C) = n (spmma.hxx:56-64);
 *      B_b = B + b*strideB (strideB = 0: one shared B), C_b = C + b*strideC (elements).
 *      fp16: v_smfmac_f32_16x16x64_f16, fp32 accumulate, one rounding to fp16. */
int sm_spmma_f16(const void* blob, const void* B, void* C, size_t m, size_t n, size_t k,
                 size_t batch, size_t strideB, size_t strideC, float alpha, float beta,
                 sm_stream_t stream);
int sm_spmma_f32(const void* blob, const float* B, float* C, size_t m, size_t n, size_t k,
                 size_t batch, size_t strideB, size_t strideC, float alpha, float beta,
                 sm_stream_t stream);

/* ---- (f-1) fused prune -> compress -> matmul: C_b = alpha * prune24_strip(A_b) * B_b + beta * C_b from the
 *      DENSE row-major A (m x k, lda, batch stride strideA) without materialising the compressed blob the
 *      reference rebuilds on every call (spmma.hxx:100-113).  Bit-identical to
 *      sm_compress24_f16 + sm_spmma_f16.  Takes k % 64 == 0, n % 8 == 0 and 16-byte aligned rows; or (round 3, the span
 *      form) any k with n % 8 == 0, n <= 128, lda == k, the batches one contiguous tall matrix that ends on a 16-byte
 *      boundary and whose 128-row span + B fit the LDS (k = 147, the ResNets' stem layer); or (round 5, the THIN form) n < 8 with
 *      k <= 64, lda == k, the batches one contiguous tall matrix of a multiple of 16 bytes with a shared B -- the depthwise
 *      convolutions of MobileNet-type networks as im2col products (n = 1, k = 9 / 25) -- computed on the vector ALUs: that form is
 *      inside the tight bound of the exact product (one rounding + k fp32 accumulation steps) but NOT bit-identical to the staged
 *      pair, whose matrix instruction adds the same products in another order.  Returns SM_STATUS_NOT_SUPPORTED otherwise (use
 *      the staged pair). */
int sm_spmma_fused_f16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                       size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha,
                       float beta, sm_stream_t stream);

/* The matmul step on `count` same-shape compressed operands as ONE grid per 8 (extension, round 4): host arrays of device
 * pointers blobs[i], B[i], C[i]; the same kernels and the same C, bit for bit, as `count` calls of sm_spmma_f16.  What the
 * grouped form buys is the same as for the fused kernels below: a few-tile layer shape's last partial round of workgroups is
 * filled by the next instance of that shape. */
int sm_spmma_f16_grouped(size_t count, const void* const* blobs, const void* const* B, void* const* C, size_t m, size_t n, size_t k,
                         size_t batch, size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream);
int sm_spmma_bf16_grouped(size_t count, const void* const* blobs, const void* const* B, void* const* C, size_t m, size_t n, size_t k,
                          size_t batch, size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream);

/* Grouped form (extension; round 3): `count` same-shape problems -- host arrays of device pointers A[i], B[i], C[i], the way
 * the reference's batched::spmm takes its As / Cs (spmm.hxx:30-33) -- in one grid per 8 problems instead of one per problem:
 * the 3-6 instances of one layer shape in a network then share the chip (the last partial round of one instance is filled by
 * the next).  Same kernels and the same C, bit for bit, as `count` calls of sm_spmma_fused_f16; C operands 16-byte aligned. */
int sm_spmma_fused_f16_grouped(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n,
                               size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                               float alpha, float beta, sm_stream_t stream);
int sm_spmma_fused_bf16_grouped(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n,
                                size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                                float alpha, float beta, sm_stream_t stream);

/* The fused entry points with a WORKSPACE (extension, round 5): the library may then run the STREAM-K form of the fused kernel on
 * the shapes whose rounds of whole 256 x 256 tiles leave compute units idle (few row tiles, long K: 196 x 512 x 4608, 784 x 512 x 1024
 * at batch 32): the launch's 64-deep stage units are cut into one equal contiguous range per compute unit, a tile that lies inside one
 * range is computed and stored exactly as without a workspace (bit-identical), a tile cut by range borders is the sum of fp32 partial
 * sums added IN A FIXED ORDER (ascending k) by the workgroup that holds its first stages -- no atomics on data: repeated runs give
 * the same bits, and the result differs from the no-workspace result only in the order of those few fp32 additions (both within the
 * fp32-accumulation bound of the exact product).  Every other shape runs exactly as sm_spmma_fused_* does.
 * workspace: sm_spmma_fused_workspace_size bytes (4 KiB of flags + one 256 KiB slot per compute unit), 16-byte aligned, its first 4 KiB
 * ZERO before the first call; every call leaves them zero again (hipGraph replays need no memset; a non-zero word 1023 afterwards
 * means a fix-up wait gave up after ~0.3 s and the result is invalid -- it cannot happen with a zeroed flag page and one call at a time per workspace).  Calls that may run concurrently
 * (different streams) need a workspace each; the grouped form's launches share one.  Nothing is allocated or synchronised inside. */
int sm_spmma_fused_workspace_size(size_t* bytes);
/* State of a workspace's flag page after the stream has drained (blocks on `stream`; a 4 KiB read-back -- outside timed regions):
 * *state = 0 clean; 1 a fix-up timed out (word 1023 set: that launch's C is invalid and a late flag may still be raised); 2 flags raised
 * without a recorded timeout (a launch still running, or a page that was never zeroed).  After 1 or 2: zero the first 4096 bytes
 * (hipMemsetAsync) before the next call on this workspace -- a dirty page makes the next launch add a stale partial, silently. */
int sm_spmma_fused_workspace_state(const void* workspace, int* state, sm_stream_t stream);
/* What the workspace entry points do with `problems` same-shape problems of `rows` x n x k (rows = m * batch when the batches share B
 * and are stacked): *takes = 1 when they run the stream-K form, and the decomposition in plan[0 .. 24]: tg, wg, groups_full, tgl,
 * wgl, slots, longest slot range (stage units), cut[0 .. 8], cutl[0 .. 8] -- row panels (256 rows of a problem, in problem order) x
 * k / 64 stage units, groups of tg panels cut into wg slot ranges at cut[] (the last group: tgl panels, wgl slots, cutl[]); a tile
 * that lies inside one slot range is bit-identical to the no-workspace result.  For tests, bench.py and schedulers; no device work.
 * Answered for beta == 0 and a 16-byte aligned C with strideC % 8 == 0 (what the dispatch's A-stationary exception for n > 256,
 * k <= 512 requires): with another beta / C alignment those shapes may run stream-K although *takes = 0 here.  plan: >= 25 unsigned. */
int sm_spmma_fused_streamk_plan(size_t rows, size_t n, size_t k, size_t problems, int* takes, unsigned* plan);
/* The dense entry points with the same workspace: the dense twin of the stream-K form (the dense GEMM the 2:4 path is measured
 * against gets the tile economy the 2:4 path gets); same workspace contract, same bit-identity statement for uncut tiles. */
int sm_gemm_rowmajor_f16_ws(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                            size_t strideB, size_t strideC, float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream);
int sm_gemm_rowmajor_bf16_ws(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                             size_t strideB, size_t strideC, float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream);
int sm_gemm_batched_f16_ws(const void* const* A_ptrs, const void* const* B_ptrs, void* const* C_ptrs, size_t m, size_t n, size_t k, size_t batch,
                           int transpose_a, int transpose_b, float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream);
int sm_spmma_fused_f16_ws(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                          size_t strideB, size_t strideC, float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream);
int sm_spmma_fused_bf16_ws(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch, size_t strideA,
                           size_t strideB, size_t strideC, float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream);
int sm_spmma_fused_f16_grouped_ws(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n,
                                  size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                                  float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream);
int sm_spmma_fused_bf16_grouped_ws(size_t count, const void* const* A, const void* const* B, void* const* C, size_t m, size_t n,
                                   size_t k, size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                                   float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream);

/* fp32 form: the STRIP rule applied to the A fragments in registers of the dense fp32 MFMA kernel (there is no fp32 sparse
 * matrix instruction).  Equals sm_gemm_rowmajor_f32 of the STRIP-pruned A bit for bit; agrees with sm_compress24_f32 +
 * sm_spmma_f32 to fp32 accumulation order.  Needs k % 32 == 0, n % 4 == 0, 16-byte aligned rows. */
int sm_spmma_fused_f32(const float* A, const float* B, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                       size_t strideA, size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream);

/* fp32 operands on the SPARSE matrix instruction (extension, round 4; csrc/spmma_f32_split.hip): the same product -- the 2:4
 * STRIP selection made on the fp32 values, mask identical to sm_prune24_f32's -- computed by v_smfmac_f32_16x16x64_bf16 on
 * exact bfloat16 splits of both operands (x = x1 + x2 + x3, 8 + 8 + 8 significand bits) with fp32 accumulation:
 *   planes = 3: a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1     |error| <= 2^-21 * sum |a| |b|  (+ fp32 accumulation)
 *   planes = 2: a1 b1 + a1 b2 + a2 b1                             |error| <= 2^-13 * sum |a| |b|
 * (north_star asks 1e-3 relative for fp32 products; cuSPARSELt, what spmma.hxx:106-114 calls, computes fp32 operands in
 * TF32 = 10 significand bits).  Not bit-identical to sm_spmma_fused_f32 -- which stays the exact form -- and several times
 * faster: bound by the HBM stream of A instead of the fp32 matrix rate.  `workspace` receives B's bfloat16 planes
 * (sm_spmma_fused_f32_split_workspace bytes, 16-byte aligned; a strided B must be packed, strideB == k * n).  Non-finite
 * operand values give NaN in the outputs they reach.  Needs k % 64 == 0, n % 8 == 0, 16-byte aligned rows of A, B and C (ldc = n);
 * a ragged k (the stem layer's 147) runs the span form when n <= 128, lda == k, B is shared and the batches are one tall contiguous
 * A (span + B's planes inside the LDS); everything else: SM_STATUS_NOT_SUPPORTED, use sm_spmma_fused_f32. */
/* sm_gemm_rowmajor_f32_split: the DENSE product C = alpha * A * B + beta * C by the same pieces (v_mfma_f32_16x16x32_bf16, same
 * workspace, same bounds with all of A's elements in the sums) -- the dense comparator the 2:4 split form is held against, and
 * 1.5-2 x faster than the fp32-MFMA sm_gemm_rowmajor_f32 in its own right. */
int sm_spmma_fused_f32_split_workspace(size_t n, size_t k, size_t batch, size_t strideB, int planes, size_t* bytes);
/* B's planes once (round 5): for a B that stays the same across calls (weights), sm_spmma_fused_f32_split_prepare splits it into `workspace`
 * (the same sm_spmma_fused_f32_split_workspace bytes) and sm_spmma_fused_f32_split_prepared multiplies from those planes: the same kernels and the
 * same C bit for bit as sm_spmma_fused_f32_split, without its per-call streaming pass over B. */
int sm_spmma_fused_f32_split_prepare(const float* B, size_t n, size_t k, size_t batch, size_t strideB, int planes, void* workspace, size_t workspace_bytes,
                                     sm_stream_t stream);
int sm_spmma_fused_f32_split_prepared(const float* A, const void* planes_workspace, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                                      size_t strideA, size_t strideB, size_t strideC, int planes, size_t workspace_bytes, float alpha, float beta,
                                      sm_stream_t stream);
int sm_spmma_fused_f32_split(const float* A, const float* B, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                             size_t strideA, size_t strideB, size_t strideC, int planes, void* workspace, size_t workspace_bytes,
                             float alpha, float beta, sm_stream_t stream);
int sm_gemm_rowmajor_f32_split(const float* A, const float* B, float* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                               size_t strideA, size_t strideB, size_t strideC, int planes, void* workspace, size_t workspace_bytes,
                               float alpha, float beta, sm_stream_t stream);

/* ---- (a5) dense batched GEMM: replaces cublas{H,S,D}gemmBatched (gemm.hxx:80-81, 133-134,
 *      186-187).  COLUMN-major, lda = m, ldb = k, ldc = m as the reference passes them;
 *      A_ptrs/B_ptrs/C_ptrs are device arrays of `batch` device pointers (examples/gemm.cu:65-90).
 *      ta / tb (gemm.hxx:33-34): SM_OP_N or SM_OP_T; a transposed operand is read as its stored k x m
 *      (n x k) form through the same leading dimension, which must cover a stored column (m >= k for
 *      ta = T, k >= n for tb = T; SM_STATUS_INVALID_VALUE otherwise, as the vendor BLAS). */
int sm_gemm_batched_f16(const void* const* A_ptrs, const void* const* B_ptrs, void* const* C_ptrs,
                        size_t m, size_t n, size_t k, size_t batch, int ta, int tb, float alpha,
                        float beta, sm_stream_t stream);
int sm_gemm_batched_f32(const float* const* A_ptrs, const float* const* B_ptrs, float* const* C_ptrs,
                        size_t m, size_t n, size_t k, size_t batch, int ta, int tb, float alpha,
                        float beta, sm_stream_t stream);
int sm_gemm_batched_f64(const double* const* A_ptrs, const double* const* B_ptrs,
                        double* const* C_ptrs, size_t m, size_t n, size_t k, size_t batch, int ta,
                        int tb, double alpha, double beta, sm_stream_t stream);

/* Dense GEMM in the layout sm_spmma_* uses (row-major, strided batch): the like-for-like dense
 * denominator for the 2:4 kernel.  No reference counterpart (the reference's only dense GEMM is
 * the column-major pointer-array one above). */
int sm_gemm_rowmajor_f16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k,
                         size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                         float alpha, float beta, sm_stream_t stream);
int sm_gemm_rowmajor_f32(const float* A, const float* B, float* C, size_t m, size_t n, size_t k,
                         size_t lda, size_t batch, size_t strideA, size_t strideB, size_t strideC,
                         float alpha, float beta, sm_stream_t stream);

/* ---- (a6) unstructured SpMM: replace cusparseSpMM on Blocked-ELL (spmm.hxx:57-67,107-110) and
 *      on strided-batch COO (spmm.hxx:164-187).  Column-major dense operands. */
int sm_spmm_bell_f32(const float* values, const uint64_t* column_indices, size_t rows, size_t cols,
                     size_t block_size, size_t ell_cols, const float* B, float* C, size_t n,
                     float alpha, float beta, sm_stream_t stream);
int sm_spmm_coo_f32(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols,
                    size_t num_batches, const int* rows, const int* cols, const float* vals,
                    const float* B, float* C, float alpha, float beta, sm_stream_t stream);

/* Blocked-ELL with a caller-provided workspace of sm_spmm_bell_workspace_size() bytes (reusable across the
 * batches of one stream): blocks are scattered into a dense A and multiplied on the fp32 matrix cores.
 * workspace == NULL behaves as sm_spmm_bell_f32 (slow gather kernel). */
int sm_spmm_bell_workspace_size(size_t rows, size_t cols, size_t* bytes /*host*/);
int sm_spmm_bell_f32_ws(const float* values, const uint64_t* column_indices, size_t rows, size_t cols,
                        size_t block_size, size_t ell_cols, const float* B, float* C, size_t n,
                        float alpha, float beta, void* workspace, sm_stream_t stream);

/* All batches of the reference's spmm() loop (spmm.hxx:90-101) in one submission: `values`, `column_indices` and `C`
 * are HOST arrays of `batch` device pointers (every A has the same rows/cols/block_size/ell_cols, B is shared --
 * exactly what the reference's driver builds).  Workspace: sm_spmm_bell_batched_workspace_size() bytes, required.
 * NOT hipGraph-capturable, unlike every other entry point: the three pointer tables are copied from the caller's HOST
 * arrays onto the stream (a synchronously staged copy from pageable memory); capture the per-matrix
 * sm_spmm_bell_f32_ws instead, or keep the host arrays alive and unchanged for the lifetime of the graph. */
int sm_spmm_bell_batched_workspace_size(size_t rows, size_t cols, size_t batch, size_t* bytes /*host*/);
int sm_spmm_bell_batched_f32(const float* const* values, const uint64_t* const* column_indices, size_t rows,
                             size_t cols, size_t block_size, size_t ell_cols, const float* B, float* const* C,
                             size_t n, size_t batch, float alpha, float beta, void* workspace, sm_stream_t stream);

/* COO with a caller-provided workspace of sm_spmm_coo_workspace_size() bytes (the reference allocates its
 * cuSPARSE buffer inside the call, spmm.hxx:183): row-sorted input runs as CSR, row-parallel, without
 * atomics (bitwise reproducible); unsorted input falls back to the atomic kernel.  workspace == NULL
 * behaves as sm_spmm_coo_f32. */
int sm_spmm_coo_workspace_size(size_t A_num_rows, size_t* bytes /*host*/);
int sm_spmm_coo_f32_ws(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols,
                       size_t num_batches, const int* rows, const int* cols, const float* vals,
                       const float* B, float* C, float alpha, float beta, void* workspace,
                       sm_stream_t stream);

/* COO, packed form (replaces the same cusparseSpMM call, spmm.hxx:164-187): the rows of A are first copied, on the
 * device and once per call, into one stream of {LDS offset, value} entries padded per row, so that the product kernel
 * (csrc/spmm.hip: spmm_csr_packed_kernel) carries no per-entry predicates, clamps or scattered stores -- the work
 * cusparseSpMM does behind its buffer (spmm.hxx:178-183).  Workspace: sm_spmm_coo_packed_workspace_size(rows, nnz) bytes
 * (16-byte aligned), its size passed back in `workspace_bytes`.  Taken when the rows are sorted (decided on the device,
 * no synchronisation; unsorted input runs the atomic kernels) and (cols + 1) * 64 bytes fit the LDS (cols <= 2559);
 * otherwise, or with
 * a workspace that is too small (then treated as sm_spmm_coo_f32_ws's, or as none), sm_spmm_coo_f32_ws /
 * sm_spmm_coo_f32 produce the same result.  Bitwise reproducible on the sorted path; hipGraph-capturable. */
int sm_spmm_coo_packed_workspace_size(size_t A_num_rows, size_t A_nnz, size_t* bytes /*host*/);
int sm_spmm_coo_f32_packed(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols,
                           size_t num_batches, const int* rows, const int* cols, const float* vals,
                           const float* B, float* C, float alpha, float beta, void* workspace,
                           size_t workspace_bytes, sm_stream_t stream);

/* Dense-MFMA form of the same product (extension; what sparsifyme::batched::strided_coo tries first since round 4, with the exact
 * form as its fallback): the batches' dense operand is scaled by a power of two and rounded once to fp16, A is scattered dense,
 * scaled and split exactly into two fp16 planes, the product runs on the fp16 matrix instruction with fp32 accumulation and the
 * inverse scales are applied to the fp32 sums: 2-3 x faster than the exact forms at 90 % sparsity.
 * Scales (round 4, computed on the device: no synchronisation): 2^x that brings the largest |b| of a ~10^6-element sample of
 * the dense operand (1024 evenly spread runs of 1024 contiguous elements) into [2^12, 2^13) and 2^y that brings the largest |a| into [2^13, 2^14) -- so the error does not
 * depend on the operands' magnitude (1e-6-sized activations or 1e+6-sized ones convert alike).
 * Error: |C - exact| <= 2^-11 * |alpha| * sum|a||b|  (one fp16 rounding of b; 4.9e-4, inside the 1e-3 this build's fp32 products are
 * held to) + fp32 accumulation + 2^-37 * max|b| * |alpha| * sum|a| (elements of the dense operand more than 2^26 below its
 * largest one lose relative precision: they are rounded to a multiple of 2^-37 of the largest).
 * Range flag: the first int of the workspace is set != 0 on the device when an element does NOT convert -- non-finite, or more
 * than 8 x the sampled maximum (|x * scale| > 65504), or duplicates of A adding up beyond the range; the matrix kernel then
 * returns without touching C, so that the caller can run an exact entry point on the untouched operands:
 * sm_spmm_coo_fast_flag() copies the flag to the host (it synchronises the stream), which is what strided_coo does.
 * Duplicates add (in an unspecified order).  Needs A_num_cols % 64 == 0 (> 0), A_num_rows % 4 == 0 (>= 8), 16-byte aligned B and
 * C and sm_spmm_coo_fast_workspace_size bytes of workspace (SM_STATUS_NOT_SUPPORTED when the sizes overflow size_t);
 * SM_STATUS_NOT_SUPPORTED otherwise (use the exact entry points).
 * Round 5 -- the SPARSE matrix instruction for the same call: with beta == 0, A_num_rows % 4 == 0 and at most 20 % of A's entries present
 * (sm_spmm_coo_fast_form says which form a call gets: where both apply, this one for A_num_cols <= 128 and for matrices of at most 256 rows), A becomes a 2:4 image (per 1 x 4 strip its first two non-zeros, scaled and split hi + lo
 * as above; a random 10 %-dense A has a third non-zero in 0.4 % of its strips -- those entries are kept as fp32 values beside the image) and
 * the product runs on v_smfmac_f32_16x16x64_f16 with the dense operand converted inside the kernel's loader: no fp16 copy of B, any
 * A_num_cols (k % 64 != 0, k % 4 != 0 included).  Same error bound.  Flag, this form: A out of range, or a 32 x 64 block of A with more than
 * 64 third / fourth non-zeros -> nothing is written; an element of B out of range -> the 128 x 128 tiles of C that read it are not written, the
 * others are (beta == 0: the exact form overwrites all of C anyway). */
int sm_spmm_coo_fast_form(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches, float beta); /* 2: sparse matrix instruction, 1: dense-MFMA pipeline, 0: not taken */
int sm_spmm_coo_fast_workspace_size(size_t A_num_rows, size_t A_num_cols, size_t B_num_cols, size_t num_batches, size_t* bytes /*host*/);
int sm_spmm_coo_fast_flag(const void* workspace, int* host_flag, sm_stream_t stream);
int sm_spmm_coo_f32_fast(size_t A_num_rows, size_t A_num_cols, size_t A_nnz, size_t B_num_cols, size_t num_batches,
                         const int* rows, const int* cols, const float* vals, const float* B, float* C, float alpha,
                         float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream);

/* ---- support: counter-based uniform fill (replaces the Thrust RNG transform of
 *      include/sparsify.me/util/gen.hxx:12-20); element i depends only on (seed, i). */
int sm_fill_uniform_f16(void* out, size_t count, uint64_t seed, float lo, float hi, sm_stream_t stream);
int sm_fill_uniform_f32(float* out, size_t count, uint64_t seed, float lo, float hi, sm_stream_t stream);

/* ---- support: plain streaming device-to-device copy (16-byte accesses; src, dst 16-byte aligned, bytes % 16 == 0).  No
 *      reference counterpart: the bandwidth yardstick bench.py times next to the step (roofline.yardstick), in-process. */
int sm_copy_bytes(const void* src, void* dst, size_t bytes, sm_stream_t stream);

/* ---- transposed operands of sparsifyme::spmma (spmma.hxx:30-31,67-69: transpose_a / transpose_b go to the vendor's matmul
 *      descriptor).  out[b][c * ld_out + r] = in[b][r * ld_in + c] for `batch` row-major rows x cols matrices of 2-, 4- or
 *      8-byte elements (out of place).  include/sparsify.me/spmma.hxx brings a transposed operand to the N form with
 *      it, prunes / compresses / multiplies there, and writes the pruned A back in its stored orientation. */
int sm_transpose(const void* in, void* out, size_t rows, size_t cols, size_t ld_in, size_t ld_out, size_t elt_bytes,
                 size_t batch, size_t stride_in, size_t stride_out, sm_stream_t stream);

/* ---- bfloat16 forms (extension; SURVEY.md 8(f) rank 2: the vendor call behind spmma.hxx:40-113 lists bf16 among its
 *      2:4 types, examples/libcusparse_lt/include/cusparseLt.h:164-169).  Same arguments, blob layout and rules as the
 *      _f16 entry points: the selection looks at magnitude bit patterns only, so prune (STRIP), check, compress and
 *      decompress ARE the fp16 kernels; the TILE rule (sums of magnitudes), the matmuls (v_smfmac_f32_16x16x64_bf16 /
 *      v_mfma_f32_16x16x32_bf16, fp32 accumulate) and the final round-to-nearest-even have their own code. */
int sm_prune24_bf16(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, int alg, sm_stream_t stream);
int sm_prune24_check_bf16(const void* A, size_t m, size_t k, size_t ld, int* d_valid, sm_stream_t stream);
int sm_compress24_bf16(const void* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob,
                       sm_stream_t stream);
int sm_decompress24_bf16(const void* blob, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* A,
                         sm_stream_t stream);
int sm_spmma_bf16(const void* blob, const void* B, void* C, size_t m, size_t n, size_t k, size_t batch,
                  size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream);
int sm_spmma_fused_bf16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                        size_t strideA, size_t strideB, size_t strideC, float alpha, float beta, sm_stream_t stream);
int sm_gemm_rowmajor_bf16(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda,
                          size_t batch, size_t strideA, size_t strideB, size_t strideC, float alpha, float beta,
                          sm_stream_t stream);
int sm_fill_uniform_bf16(void* out, size_t count, uint64_t seed, float lo, float hi, sm_stream_t stream);

/* ---- int8 forms (extension; SURVEY.md 8(f) rank 2, cusparseLt.h:164-169).  Elements are signed bytes; the rules
 *      act on |x| (|-128| = 128); same blob geometry with 1-byte elements (sm_compress24_size(m, k, 1, batch, ...)).
 *      sm_spmma_i8: C (int32, row-major m x n) = A_2:4 . B (+ C when
 *      accumulate != 0), exact integer arithmetic on v_smfmac_i32_16x16x128_i8; B is [n][k] -- K-CONTIGUOUS per output
 *      column ("TN", the layout int8 matrix cores are fed in), B_b = B + b * strideB (0 = shared).  Needs k % 64 == 0,
 *      an even m, a 16-byte aligned B; SM_STATUS_NOT_SUPPORTED otherwise. */
/* out[c][r] = in[r][c] for a row-major rows x cols byte matrix (out of place): turns the reference's row-major k x n B
 * (spmma.hxx:40-64) into the [n][k] operand of sm_spmma_i8; a one-off for the small, reused weight operand */
int sm_transpose_i8(const void* in, void* out, size_t rows, size_t cols, sm_stream_t stream);
int sm_prune24_i8(const void* A_in, void* A_out, size_t m, size_t k, size_t ld, int alg, sm_stream_t stream);
int sm_prune24_check_i8(const void* A, size_t m, size_t k, size_t ld, int* d_valid, sm_stream_t stream);
int sm_compress24_i8(const void* A, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* blob,
                     sm_stream_t stream);
int sm_decompress24_i8(const void* blob, size_t m, size_t k, size_t ld, size_t batch, size_t strideA, void* A,
                       sm_stream_t stream);
int sm_spmma_i8(const void* blob, const void* B, int32_t* C, size_t m, size_t n, size_t k, size_t batch,
                size_t strideB, size_t strideC, int accumulate, sm_stream_t stream);
/* the same product requantised on the way out: C (int8) = saturate(round_to_nearest_even(scale * acc)), one fp32
 * multiply of the int32 accumulator converted to fp32 */
int sm_spmma_i8_q(const void* blob, const void* B, void* C, size_t m, size_t n, size_t k, size_t batch, size_t strideB,
                  size_t strideC, float scale, sm_stream_t stream);
/* prune (STRIP) + compress + matmul in one kernel straight from the dense int8 A (row-major, lda): bit-identical to
 * sm_compress24_i8 + sm_spmma_i8[_q], no blob.  Needs k % 64 == 0 and 16-byte aligned rows; SM_STATUS_NOT_SUPPORTED
 * otherwise (use the pair). */
int sm_spmma_fused_i8(const void* A, const void* B, int32_t* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                      size_t strideA, size_t strideB, size_t strideC, int accumulate, sm_stream_t stream);
int sm_spmma_fused_i8_q(const void* A, const void* B, void* C, size_t m, size_t n, size_t k, size_t lda, size_t batch,
                        size_t strideA, size_t strideB, size_t strideC, float scale, sm_stream_t stream);

/* ---- im2col front end (extension).  X: N x C x H x W activations (NCHW, contiguous).
 *      A: per image the row-major L x K operand of the layer's matmul, L = out_h * out_w rows (row oh * out_w + ow),
 *      K = C * kh * kw columns (column c * kh * kw + r * kw + u), images back to back -- the transpose of torch's
 *      unfold, i.e. the (m, k) = (L, C * kh * kw) operand of the reference's shape tables
 *      (datasets/get_shapes.py:30-40, 66-73).  out = floor((in + 2 pad - dilation (k - 1) - 1) / stride) + 1
 *      (get_shapes.py:19-20).  sm_im2col_compress24_* writes sm_compress24_*'s blob of that A (m = L, k = K,
 *      batch = N; size from sm_compress24_size) without ever materialising the dense A; same bytes as
 *      sm_im2col_* followed by sm_compress24_*. */
int sm_conv_out_size(size_t in, size_t kernel, size_t stride, size_t pad, size_t dilation, size_t* out /*host*/);
int sm_im2col_f16(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad,
                  size_t dilation, void* A, sm_stream_t stream);
int sm_im2col_bf16(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad,
                   size_t dilation, void* A, sm_stream_t stream);
int sm_im2col_compress24_f16(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                             size_t pad, size_t dilation, void* blob, sm_stream_t stream);
int sm_im2col_compress24_bf16(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                              size_t pad, size_t dilation, void* blob, sm_stream_t stream);

/* ---- implicit-GEMM form for convolution layers (extension; SURVEY.md 8(f) rank 3, datasets/get_shapes.py:30-40,66-73):
 *      C[i][l][:] = alpha * prune24_strip(A_i)[l][:] * B + beta * C[i][l][:] with A_i = the im2col operand of image i
 *      (sm_im2col_*'s layout: L = out_h * out_w rows, K = Cin * kh * kw columns), computed straight from the NCHW
 *      activations X: neither the dense A nor its blob is ever written to or read from HBM.  B: K x n_out row-major
 *      (shared by the images), C: N * L rows x n_out row-major.  Bit-identical to sm_im2col_compress24_* followed by
 *      sm_spmma_* (m = L, k = K, batch = N, strideB = 0).  Needs K % 64 == 0, n_out % 8 == 0, an even W of at most ~120
 *      columns and kh * kw <= 64; SM_STATUS_NOT_SUPPORTED otherwise (use the pair). */
int sm_conv_spmma_fused_f16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw,
                            size_t stride, size_t pad, size_t dilation, size_t n_out, float alpha, float beta, sm_stream_t stream);
int sm_conv_spmma_fused_bf16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw,
                             size_t stride, size_t pad, size_t dilation, size_t n_out, float alpha, float beta, sm_stream_t stream);
/* The same product by the faster of its two routes (round 4): the implicit-GEMM kernel, or -- small-spatial layers with a long K
 * (out_h * out_w <= 256 and K >= 2048: the 14 x 14 x 512-channel layers of a ResNet) -- sm_im2col_compress24_* into `workspace`
 * followed by sm_spmma_*.  Same C bit for bit either way.  sm_conv_spmma_workspace gives the bytes: the blob's size
 * (sm_compress24_size(out_h * out_w, Cin * kh * kw, 2, N)) for the layers the rule sends to the pair AND for every geometry the
 * implicit kernel cannot run (K % 64 != 0 such as a 7 x 7 x 3 stem, an odd or too wide W, kh * kw > 64, a patch beyond its DMA /
 * LDS limits), so that a caller who sizes the workspace with it never gets SM_STATUS_NOT_SUPPORTED for a geometry reason; 0 where
 * the implicit kernel is kept.  The query sees neither n_out nor the pointers: for n_out % 8 != 0 or a B / X that is not 16- /
 * 4-byte aligned size the workspace with sm_compress24_size.  With no workspace the implicit kernel runs wherever it can. */
int sm_conv_spmma_workspace(size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad, size_t dilation,
                            size_t* bytes);
int sm_conv_spmma_f16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                      size_t pad, size_t dilation, size_t n_out, float alpha, float beta, void* workspace, size_t workspace_bytes,
                      sm_stream_t stream);
int sm_conv_spmma_bf16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                       size_t pad, size_t dilation, size_t n_out, float alpha, float beta, void* workspace, size_t workspace_bytes,
                       sm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSIFYME_H_ */
