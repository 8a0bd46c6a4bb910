// im2col.hip -- the front end of the hot path: NCHW activations -> the per-image row-major
// operand A[L x K] of the layer's matmul (L = out_h * out_w rows, K = C * kh * kw columns, column index
// c * kh * kw + r * kw + u), i.e. the transpose of what torch's unfold returns and what the reference's shape walk
// multiplies (datasets/get_shapes.py:30-40: m = L, k = C * kh * kw).  Two forms, same kernel:
//   sm_im2col_{f16,bf16}             writes the dense A (K-contiguous rows, batch images back to back);
//   sm_im2col_compress24_{f16,bf16}  applies the 2:4 STRIP selection to each row on the way out and writes the
//                                    compressed blob sm_spmma_* consumes: the 9x (3x3) or 49x (7x7) larger dense A is
//                                    never written to or read from HBM -- bit-identical to sm_compress24 of the former.
// One workgroup = one image row of outputs (oh) x as many output columns as keep a window row's input span within 64
// columns (one wave instruction) x one chunk of channels (blockIdx.y: chunks own disjoint column ranges of A).  The
// input patch [channels][kh][input columns] is staged in LDS (eight segments' loads in flight per wave; segment pitch
// an odd number of dwords: at a pitch of 64 elements all segments start in bank 0 and the emit phase ran 2x slower),
// then every thread emits items of 8 consecutive k of one output row: one 16-byte read of a per-workgroup table of
// patch offsets, 8 LDS reads, one 16-byte store (or 8 bytes of kept values + 1 metadata byte).  A 1 x 1 / stride 1 /
// no-padding window is a transpose of [C][H * W] and runs with the image flattened to one row.  First version:
// 1.4-3.7 TB/s on the ResNet-50 convolutions at batch 32, bound by the emit loop's instructions
// (profiles/im2col_probe_r01.txt); the fused form takes 2/3 of the time of im2col followed by compress.
#include "select24.h"
#include "sm_common.h"

namespace sm {

struct Im2colArgs {
  const uint16_t* X;
  uint16_t* A;          // dense output, or the blob's values section
  unsigned char* meta;  // blob metadata (compress form)
  int N, C, H, W, kh, kw, stride, pad, dil;
  int OH, OW, K, kc, CB, IWB, PIT, owblocks, owb;  // PIT: LDS pitch of a patch segment (elements; odd dword count)  // owb: output columns per workgroup, chosen so that IWB <= 64
  size_t M;             // N * OH * OW blob rows
};


template <bool COMPRESS>
__global__ __launch_bounds__(256) void im2col16_kernel(const Im2colArgs p) {
  // [CB][kh][PIT] patch, then the offset table: entry i = patch offset of chunk-relative column i at output column 0
  extern __shared__ __attribute__((aligned(16))) uint16_t patch[];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  unsigned bid = blockIdx.x;
  const int owb = (int)(bid % (unsigned)p.owblocks);
  bid /= (unsigned)p.owblocks;
  const int oh = (int)(bid % (unsigned)p.OH), n = (int)(bid / (unsigned)p.OH);
  const int ow0 = owb * p.owb, nrows = p.OW - ow0 < p.owb ? p.OW - ow0 : p.owb;
  const int iw0 = ow0 * p.stride - p.pad, ih0 = oh * p.stride - p.pad;
  const int khkw = p.kh * p.kw, L = p.OH * p.OW;
  const bool vec_ok = !COMPRESS && (p.K % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15u) == 0);
  uint16_t* tab = patch + p.CB * p.kh * p.PIT;  // 16-byte aligned: CB is a multiple of 8
  const int tabn = p.CB * khkw;                 // a multiple of 8
  for (int i = (int)tid; i < tabn; i += 256) {
    const int c = i / khkw, rem = i - c * khkw, r = rem / p.kw, u = rem - r * p.kw;
    tab[i] = (uint16_t)((c * p.kh + r) * p.PIT + u * p.dil);
  }
  const size_t Rbase = (size_t)n * L + (size_t)oh * p.OW + ow0;

  {  // one channel chunk per workgroup (blockIdx.y): chunks own disjoint column ranges of the output
    const int c0 = (int)blockIdx.y * p.CB;
    const int cb = p.C - c0 < p.CB ? p.C - c0 : p.CB;
    // ---- stage the patch: one (channel, window row) segment of IWB <= 64 input columns per wave instruction, eight
    //      segments' loads in flight per wave before the first LDS store
    const int nseg = cb * p.kh;
    const int iw = iw0 + (int)lane;
    const bool col_ok = (int)lane < p.IWB && iw >= 0 && iw < p.W;
    const uint16_t* Xn = p.X + ((size_t)n * p.C + c0) * p.H * p.W + iw;  // this lane's column of channel c0, row 0
    const size_t chan = (size_t)p.H * p.W;
    for (int sb = 8 * (int)wave; sb < nseg; sb += 32) {
      uint16_t v[8];
      int c = sb / p.kh, r = sb - c * p.kh;  // wave-uniform; stepped through the 8 segments
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int ih = ih0 + r * p.dil;
        const bool ok = sb + t < nseg && col_ok && ih >= 0 && ih < p.H;
        v[t] = ok ? Xn[(size_t)c * chan + (size_t)ih * p.W] : (uint16_t)0;
        if (++r == p.kh) { r = 0; ++c; }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (sb + t < nseg && (int)lane < p.IWB) patch[(sb + t) * p.PIT + lane] = v[t];
    }
    __syncthreads();
    // ---- emit: items of 8 consecutive k of one output row; the last chunk of the compress form runs on to kc
    const int k0 = c0 * khkw;  // a multiple of 8: CB is
    const bool last = c0 + cb >= p.C;
    const int kend = last ? (COMPRESS ? p.kc : p.K) : (c0 + cb) * khkw;
    // item fastest, then row: a row's items are contiguous in the dense A, and runs of 64 bytes in the blob (ordering
    // the compressed form row-fastest within a stage, i.e. fully contiguous stores, measured 7 % slower: the kernel is
    // bound by the emit loop's instructions, not by its store pattern)
    const int nitems = (kend - k0 + 7) / 8, total = nrows * nitems;
    const int dj = 256 / nitems, dq = 256 - dj * nitems;  // the (row, item) step of a 256-thread stride
    int j = (int)tid / nitems, q = (int)tid - j * nitems;
    for (int w = (int)tid; w < total; w += 256, j += dj, q += dq) {
      if (q >= nitems) { q -= nitems; ++j; }
      const int kr = 8 * q, k = k0 + kr;  // chunk-relative / absolute first column of the item
      __attribute__((aligned(16))) uint16_t e[8];
      const int js = j * p.stride;
      if (k + 8 <= p.K) {
        const u4 o4 = *reinterpret_cast<const u4*>(tab + kr);
        const uint16_t* o = reinterpret_cast<const uint16_t*>(&o4);
#pragma unroll
        for (int t = 0; t < 8; ++t) e[t] = patch[o[t] + js];
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) e[t] = (k + t < p.K) ? patch[tab[kr + t] + js] : (uint16_t)0;
      }
      const size_t R = Rbase + (size_t)j;
      if constexpr (COMPRESS) {
        const u4 d = *reinterpret_cast<const u4*>(e);
        uint32_t kept0, kept1, n0, n1;
        strip_select_f16(d[0], d[1], kept0, n0);  // a strip at or beyond K is all zeros: keeps (0, 1), nibble 0x4
        strip_select_f16(d[2], d[3], kept1, n1);
        const size_t itp = ((((size_t)(k >> 6)) * p.M + R) << 3) + (size_t)((k & 63) >> 3);
        *reinterpret_cast<u2*>(p.A + itp * 4) = u2{kept0, kept1};
        p.meta[itp] = (unsigned char)(n0 | (n1 << 4));
      } else {
        uint16_t* dst = p.A + R * (size_t)p.K + k;
        if (vec_ok) {
          __builtin_nontemporal_store(*reinterpret_cast<const u4*>(e), reinterpret_cast<u4*>(dst));
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (k + t < p.K) dst[t] = e[t];
        }
      }
    }
  }
}

static bool conv_out(size_t in, size_t k, size_t stride, size_t pad, size_t dil, size_t* out) {
  if (k == 0 || stride == 0 || dil == 0) return false;
  const size_t span = dil * (k - 1) + 1;
  if (in + 2 * pad < span) return false;
  *out = (in + 2 * pad - span) / stride + 1;
  return true;
}

static int launch_im2col(bool compress, const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw,
                         size_t stride, size_t pad, size_t dil, void* out, hipStream_t st) {
  size_t OH = 0, OW = 0;
  if (!X || !out || !conv_out(H, kh, stride, pad, dil, &OH) || !conv_out(W, kw, stride, pad, dil, &OW)) {
    set_error("sm_im2col: invalid argument (null pointer, zero kernel / stride / dilation, or kernel larger than the padded input)");
    return SM_STATUS_INVALID_VALUE;
  }
  if (N == 0 || C == 0) return SM_STATUS_SUCCESS;
  const size_t K = C * kh * kw, L = OH * OW;
  // a 1 x 1 window at stride 1 without padding is a transpose of [C][H * W]: treat the image as one long row, so that
  // a workgroup's output columns are 64 consecutive pixels whatever the image width
  if (kh == 1 && kw == 1 && stride == 1 && pad == 0) {
    W = H * W; H = 1; OW = L; OH = 1;
  }
  // output columns per workgroup: as many as keep the input span of a window row within one wave (64 columns)
  if ((kw - 1) * dil + 1 > 64) {
    set_error("sm_im2col: window row of %zu columns (dilation %zu) not supported", kw, dil);
    return SM_STATUS_NOT_SUPPORTED;
  }
  const size_t OWB = (64 - ((kw - 1) * dil + 1)) / stride + 1;
  const size_t owblocks = ceil_div(OW, OWB);
  const size_t IWB = (OWB - 1) * stride + (kw - 1) * dil + 1;
  // LDS pitch of a segment: an odd number of dwords, so that the segments a wave's lanes read in the emit phase (same
  // column, different channel / window row) fall into different banks (at a pitch of 64 elements all would share one)
  size_t PIT = (IWB + 1) / 2 * 2;
  if ((PIT / 2) % 2 == 0) PIT += 2;
  const size_t budget = 8192;  // 16 KiB of 16-bit patch per workgroup: several workgroups per CU hide the staging latency
  if (kh * PIT * 8 > budget || K > 0x7fffffffull || N * OH * owblocks > 0x7fffffffull || N * L > 0x7fffffffull ||
      H > 0x3fffffffull || W > 0x3fffffffull || pad > 0x3fffffffull) {
    set_error("sm_im2col: window %zu x %zu (stride %zu, dilation %zu) or problem size not supported", kh, kw, stride, dil);
    return SM_STATUS_NOT_SUPPORTED;
  }
  size_t CB = budget / (kh * PIT) / 8 * 8;
  if (CB > ((C + 7) / 8) * 8) CB = ((C + 7) / 8) * 8;
  if (ceil_div(C, CB) > 65535) {
    set_error("sm_im2col: too many channel chunks");
    return SM_STATUS_NOT_SUPPORTED;
  }
  Im2colArgs a = {};
  a.X = (const uint16_t*)X;
  a.N = (int)N; a.C = (int)C; a.H = (int)H; a.W = (int)W; a.kh = (int)kh; a.kw = (int)kw;
  a.stride = (int)stride; a.pad = (int)pad; a.dil = (int)dil;
  a.OH = (int)OH; a.OW = (int)OW; a.K = (int)K; a.CB = (int)CB; a.IWB = (int)IWB; a.PIT = (int)PIT; a.owblocks = (int)owblocks; a.owb = (int)OWB;
  a.M = N * L;
  const size_t lds = (CB * kh * PIT + CB * kh * kw) * sizeof(uint16_t);  // patch + offset table
  const dim3 grid((unsigned)(N * OH * owblocks), (unsigned)ceil_div(C, CB));
  if (compress) {
    if (!aligned16(out)) {
      set_error("sm_im2col_compress24: the blob must be 16-byte aligned");
      return SM_STATUS_INVALID_VALUE;
    }
    const BlobLayout B = blob_layout(L, K, 2, N);
    a.kc = (int)B.kc;
    a.A = (uint16_t*)out;
    a.meta = (unsigned char*)out + B.meta_off;
    // zero the alignment gaps so that a blob is a pure function of its input (as sm_compress24 does)
    const size_t vbytes = B.M * (B.kc / 2) * 2, mbytes = B.M * (B.kc / 8);
    if (B.meta_off > vbytes && hipMemsetAsync((char*)out + vbytes, 0, B.meta_off - vbytes, st) != hipSuccess)
      return check_launch("hipMemsetAsync");
    if (B.total > B.meta_off + mbytes &&
        hipMemsetAsync((char*)out + B.meta_off + mbytes, 0, B.total - B.meta_off - mbytes, st) != hipSuccess)
      return check_launch("hipMemsetAsync");
    im2col16_kernel<true><<<grid, 256, lds, st>>>(a);
  } else {
    a.kc = (int)K;
    a.A = (uint16_t*)out;
    im2col16_kernel<false><<<grid, 256, lds, st>>>(a);
  }
  return check_launch("im2col16_kernel");
}

}  // namespace sm

using namespace sm;

extern "C" {

int sm_conv_out_size(size_t in, size_t kernel, size_t stride, size_t pad, size_t dilation, size_t* out) {
  if (!out || !conv_out(in, kernel, stride, pad, dilation, out)) {
    set_error("sm_conv_out_size: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  return SM_STATUS_SUCCESS;
}
int sm_im2col_f16(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad,
                  size_t dilation, void* A, sm_stream_t s) {
  return launch_im2col(false, X, N, C, H, W, kh, kw, stride, pad, dilation, A, (hipStream_t)s);
}
int sm_im2col_bf16(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad,
                   size_t dilation, void* A, sm_stream_t s) {
  return launch_im2col(false, X, N, C, H, W, kh, kw, stride, pad, dilation, A, (hipStream_t)s);
}
int sm_im2col_compress24_f16(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                             size_t pad, size_t dilation, void* blob, sm_stream_t s) {
  return launch_im2col(true, X, N, C, H, W, kh, kw, stride, pad, dilation, blob, (hipStream_t)s);
}
int sm_im2col_compress24_bf16(const void* X, size_t N, size_t C, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                              size_t pad, size_t dilation, void* blob, sm_stream_t s) {
  return launch_im2col(true, X, N, C, H, W, kh, kw, stride, pad, dilation, blob, (hipStream_t)s);
}

}  // extern "C"
