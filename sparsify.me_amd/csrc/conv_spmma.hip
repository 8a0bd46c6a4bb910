// conv_spmma.hip -- implicit-GEMM form of the hot path for convolution layers ("fuse im2col into
// the A-tile loader so A is never materialised"; reference datasets/get_shapes.py:30-40,66-73 defines the operand:
// A = unfold(X) transposed, m = L = out_h * out_w rows per image, k = C * kh * kw columns, column c * kh * kw + r * kw + u).
//   C[n][l][:] = alpha * prune24_strip(A_n)[l][:] * B + beta * C[n][l][:]
// straight from the NCHW activations X: neither the kh*kw-times larger dense A nor its 2:4 blob ever exists in HBM.
// Bit-identical to sm_im2col_compress24_* followed by sm_spmma_* (same STRIP selection on the same values, the same
// v_smfmac sequence per output element).
//
// Structure = the direct fused kernel (spmma_f16_fused.hip) with the dense-A stage replaced by an activation PATCH:
//   workgroup = 128 consecutive output pixels of one image x BN output channels, 4 waves, wave w owns pixels 32w..32w+31
//   and all BN columns; K advances in stages of 64 (c, r, u) columns = `nch` whole or partial input channels;
//   per stage the workgroup brings, by LDS-DMA (4 bytes per lane, several patch rows per wave instruction), the input
//   rows its pixels' windows touch for those channels into LDS -- [channel][input row][padl + W] halves, zero borders
//   kept in place so the gather needs no bounds checks -- and the B tile [64][BN] as the other matmul kernels do;
//   the lane that feeds row l, k-group g to the SMFMAC gathers its 16 values (2-byte LDS reads at
//   patch + pixel_offset(l) + k_offset(k); the k offsets come from a table with one 64-entry row per phase of
//   (64 * stage) mod (kh * kw), built once per workgroup), applies the 2:4 STRIP selection in registers and issues the
//   SMFMACs.  Double-buffered stages, one barrier per stage.
// HBM bytes: X once per column tile (+ halo re-reads served by L2) + B + C, instead of kh*kw times X.
#include "mma_tile.h"

namespace sm {

struct ConvArgs {
  const half_t* X;
  const half_t* B;
  half_t* C;
  int N, Cin, H, W, kh, kw, stride, pad, dil, OH, OW, L, Nout, K, khkw;
  int tiles_m, tiles_n;
  int RI;      // patch rows per channel: (max output-row span of a tile) * stride + (kh - 1) * dil + 1
  int pitch;   // patch row pitch in halves = padl + W (even); a row's right border is the next row's left border
  int padl;    // zero halves in front of each row (even, >= pad and >= what (kw - 1) * dil - pad overshoots)
  int rpi;     // patch rows per DMA wave-instruction = 64 / (pitch / 2)
  int nch;     // input channels a 64-k stage can touch
  int a_n;     // patch DMA instructions per stage
  int patch_bytes;  // per stage buffer, 16-byte multiple (incl. the zero tail after the last row)
  float alpha, beta;
  int ablate;  // diagnostic timing builds (-DSM_TUNING, SM_CONV_ABLATE): 1 no patch DMA, 2 no gather, 4 no B DMA, 8 no SMFMAC; 0 in the product
};

__device__ __attribute__((aligned(256))) const unsigned char sm_conv_zero_page[256] = {0};

// V16 (round 4): 16-byte patch DMAs where the input rows are whole 16-byte pieces (W % 8 == 0: the 112- and 56-wide layers) --
// a lane moves 8 halves, the row pitch is padl + W with padl a multiple of 8, so one wave instruction brings 4 (W = 112) or 8
// (W = 56) patch rows instead of 1 or 2 and a stage needs a quarter of the DMA instructions (and of the plan's registers).
// SMALL: the stage's patch needs at most 16 DMA instructions (always with V16; with 4-byte pieces for W <= 30): the plan then
// holds 4 slots per wave instead of 12 and the kernel is asked to fit four (64-column tiles) or three (128) waves per SIMD --
// it does, without scratch (102 / 142 registers against 167 / 238), so that three to four workgroups share a CU as far as
// the LDS allows: the "fourth wave per SIMD" lever of DESIGN.md 4.4.
template <int BN, bool BF, bool V16 = false, bool SMALL = V16>
__global__ __launch_bounds__(256, SMALL ? (BN == 64 ? 4 : 3) : 1) void conv_spmma_fused_kernel(const ConvArgs p) {
  constexpr int BM = 128, NW = 4, TM = 32, FM = 2, FN = BN / 16;
  constexpr int SB = 64 * BN * 2, B_N = BN / 8;
  constexpr int EPL = V16 ? 8 : 2;    // halves per lane of a patch DMA
  static_assert(SMALL || !V16, "16-byte pieces imply the small plan");
  constexpr int MAXA = SMALL ? 4 : 12;  // patch DMA slots per wave (a_n <= 16 / 48)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int STAGE = p.patch_bytes + SB;
  // layout: [stage 0: patch | B][stage 1: patch | B][k-offset table: khkw rows of 64 u16]
  char* tab = smem + 2 * STAGE;

  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned tiles = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned img = lid / tiles, trem = lid - img * tiles;
  const unsigned tile_m = trem / (unsigned)p.tiles_n, tile_n = trem - tile_m * (unsigned)p.tiles_n;
  const int m0 = (int)tile_m * BM, n0 = (int)tile_n * BN;
  const int nkt = p.K / 64;
  const int plast = (m0 + BM - 1 < p.L ? m0 + BM - 1 : p.L - 1);
  const int oh_first = m0 / p.OW;
  const int ih_lo = oh_first * p.stride - p.pad;  // input row of patch row 0
  (void)plast;

  // ---- once: zero both patch buffers (borders stay zero for the whole kernel: the DMA only writes the W data
  //      columns of a row), build the k-offset table
  for (unsigned o = tid * 16u; o < (unsigned)p.patch_bytes; o += 256u * 16u) {
    *reinterpret_cast<u4*>(smem + o) = u4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u4*>(smem + STAGE + o) = u4{0u, 0u, 0u, 0u};
  }
  for (int e = (int)tid; e < p.khkw * 64; e += 256) {
    const int phase = e >> 6, kk = e & 63;
    const int q = phase + kk;                 // column index relative to the stage's first channel
    const int c = q / p.khkw, rem = q - c * p.khkw, r3 = rem / p.kw, u = rem - r3 * p.kw;
    reinterpret_cast<unsigned short*>(tab)[e] = (unsigned short)(((c * p.RI + r3 * p.dil) * p.pitch + u * p.dil) * 2);
  }

  // ---- per-lane DMA plan (fixed over the stages except for the channel base)
  const int pdw = p.pitch / EPL;  // lanes per patch row
  const size_t chan_bytes = (size_t)p.H * p.W * 2;
  const char* Ximg = reinterpret_cast<const char*>(p.X) + (size_t)img * p.Cin * chan_bytes;
  long long a_src[MAXA];   // byte offset from Ximg + cbase * chan_bytes; < 0: this lane reads zeros
  int a_ch[MAXA];          // channel (relative) of the lane's row, for the Cin bound
  bool a_on[MAXA];         // lane takes part in slot i
#pragma unroll
  for (int i = 0; i < MAXA; ++i) {
    const int t = (int)wave + NW * i;
    const int prow = t * p.rpi + (int)lane / pdw, d = (int)lane % pdw;
    const int ch = prow / p.RI, rr = prow - ch * p.RI, ih = ih_lo + rr;
    a_on[i] = t < p.a_n && (int)lane < p.rpi * pdw && prow < p.nch * p.RI && d >= p.padl / EPL && d < p.padl / EPL + p.W / EPL;
    a_ch[i] = ch;
    a_src[i] = (ih >= 0 && ih < p.H) ? (long long)(((size_t)ch * p.H + ih) * p.W * 2 + (size_t)(d - p.padl / EPL) * (EPL * 2)) : -1;
  }
  // B slots: B_N instructions of 8 k-rows x 128 B, instruction j -> wave j % 4
  constexpr int SLB = (B_N + NW - 1) / NW;
  const char* b_src[SLB];
  unsigned b_lds[SLB];
#pragma unroll
  for (int i = 0; i < SLB; ++i) {
    const unsigned j = wave + (unsigned)NW * i, panel = j >> 3, kr = 8u * (j & 7u) + (lane >> 3);
    const unsigned cs = (lane & 7u) ^ b_swz(kr);
    int gc = n0 + (int)(64u * panel + 8u * cs);
    gc = gc <= p.Nout - 8 ? gc : p.Nout - 8;
    b_src[i] = reinterpret_cast<const char*>(p.B + (size_t)kr * p.Nout + gc);
    b_lds[i] = panel * 8192u + (j & 7u) * 1024u;
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
    const int cbase = (64 * kt) / p.khkw;
    const char* Xc = Ximg + (size_t)cbase * chan_bytes;
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
      const int t = (int)wave + NW * i;  // wave-uniform
      if (t >= p.a_n) break;
      if (a_on[i] && !(p.ablate & 1)) {
        const bool live = a_src[i] >= 0 && cbase + a_ch[i] < p.Cin;
        if constexpr (V16) {
          gptr_t* g = live ? (gptr_t*)(Xc + a_src[i]) : (gptr_t*)(sm_conv_zero_page + 16u * (lane & 15u));
          __builtin_amdgcn_global_load_lds(g, (lptr_t*)(base + (size_t)t * p.rpi * pdw * 16), 16, 0, 0);
        } else {
          gptr_t* g = live ? (gptr_t*)(Xc + a_src[i]) : (gptr_t*)(sm_conv_zero_page + 4u * lane);
          __builtin_amdgcn_global_load_lds(g, (lptr_t*)(base + (size_t)t * p.rpi * pdw * 4), 4, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < SLB; ++i) {
      const unsigned j = wave + (unsigned)NW * i;
      if (j >= (unsigned)B_N) break;
      if (p.ablate & 4) continue;
      __builtin_amdgcn_global_load_lds((gptr_t*)(b_src[i] + (size_t)kt * 64 * p.Nout * 2), (lptr_t*)(base + p.patch_bytes + b_lds[i]), 16, 0, 0);
    }
  };

  // ---- per-lane pixel offsets of the two 16-row fragments
  const unsigned g = lane >> 4, r16 = lane & 15u;
  unsigned poff[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int px = m0 + (int)(wave * TM + i * 16 + r16);
    px = px < p.L ? px : p.L - 1;  // rows past the image: any valid pixel (their products are never stored)
    const int oh = px / p.OW, ow = px - oh * p.OW;
    poff[i] = (unsigned)((((oh - oh_first) * p.stride) * p.pitch + ow * p.stride + p.padl - p.pad) * 2);
  }

  f4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();  // zeroed patches and the table are in place before any DMA lands / any gather reads
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    wait_dma_and_barrier<0>();  // stage kt landed for every wave; every wave finished reading the other buffer
    if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
    const char* Ps = smem + (kt & 1) * STAGE;
    const char* Bs = Ps + p.patch_bytes;
    const int phase = (64 * kt) % p.khkw;
    const u4 t0 = *reinterpret_cast<const u4*>(tab + phase * 128 + 32 * g);
    const u4 t1 = *reinterpret_cast<const u4*>(tab + phase * 128 + 32 * g + 16);
    const uint32_t tw[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
    h8 af[FM];
    int idx[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const char* base = Ps + poff[i];
      uint32_t d[8];
      if (p.ablate & 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = tw[e] + poff[i];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t lo = *reinterpret_cast<const unsigned short*>(base + (tw[e] & 0xffffu));
          const uint32_t hi = *reinterpret_cast<const unsigned short*>(base + (tw[e] >> 16));
          d[e] = lo | (hi << 16);
        }
      }
      uint32_t k0, k1, k2, k3, q0, q1, q2, q3;
      strip_select_f16(d[0], d[1], k0, q0);
      strip_select_f16(d[2], d[3], k1, q1);
      strip_select_f16(d[4], d[5], k2, q2);
      strip_select_f16(d[6], d[7], k3, q3);
      af[i] = __builtin_bit_cast(h8, u4{k0, k1, k2, k3});
      idx[i] = (int)(q0 | (q1 << 4) | (q2 << 8) | (q3 << 12));
    }
    // B fragments by the transposing read, fragment j+1's reads in flight under fragment j's SMFMACs (mma_tile.h)
    const unsigned bs_addr = (unsigned)(uintptr_t)(lds_char*)Bs;
    s4 x0[2], x1[2], x2[2], x3[2];
    auto issue = [&](int j, s4& v0, s4& v1, s4& v2, s4& v3) {
      const unsigned c0 = j * 16, q = r16 >> 2, pp = r16 & 3u;
      const unsigned a = bs_addr + b_off<64>(8u * g + q, c0 + 4u * pp);
      asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                   "ds_read_b64_tr_b16 %2, %4 offset:4096\n\tds_read_b64_tr_b16 %3, %4 offset:4608"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a) : "memory");
    };
    issue(0, x0[0], x1[0], x2[0], x3[0]);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int c = j & 1, nx = c ^ 1;
      if (j + 1 < FN) {
        issue(j + 1, x0[nx], x1[nx], x2[nx], x3[nx]);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x0[c]), "+v"(x1[c]), "+v"(x2[c]), "+v"(x3[c]) :: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0[c]), "+v"(x1[c]), "+v"(x2[c]), "+v"(x3[c]) :: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      typedef short s16 __attribute__((ext_vector_type(16)));
      const s16 all = {x0[c][0], x0[c][1], x0[c][2], x0[c][3], x1[c][0], x1[c][1], x1[c][2], x1[c][3],
                       x2[c][0], x2[c][1], x2[c][2], x2[c][3], x3[c][0], x3[c][1], x3[c][2], x3[c][3]};
      const h16 bf = __builtin_bit_cast(h16, all);
      if (!(p.ablate & 8))
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = smfmac16<BF>(af[i], bf, acc[i][j], idx[i]);
    }
  }
  __syncthreads();  // nothing is in flight: the last iteration issued no DMA
  half_t* C = p.C + (size_t)img * p.L * p.Nout;
  store_c_tile<BM, BN, FM, FN, 64 * NW, BF>(smem, C, acc, true, wave * TM, 0, m0, n0, p.L, p.Nout, p.alpha, p.beta, tid);
}

template <int BN, bool BF, bool V16 = false, bool SMALL = V16>
static int launch_conv(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  a.tiles_m = (a.L + 127) / 128;
  a.tiles_n = (a.Nout + BN - 1) / BN;
  const size_t nwg = (size_t)a.tiles_m * a.tiles_n * a.N;
  if (nwg == 0) return SM_STATUS_SUCCESS;
  if (nwg > 0x7fffffffu) {
    set_error("sm_conv_spmma_fused: grid too large");
    return SM_STATUS_NOT_SUPPORTED;
  }
  const size_t lds_main = 2 * ((size_t)a.patch_bytes + 64 * BN * 2) + (size_t)a.khkw * 128;
  const size_t lds_epi = (size_t)128 * (BN * 2 + 16);
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  if (lds > 160 * 1024) {
    set_error("sm_conv_spmma_fused: the activation patch of a stage (%d bytes) does not fit LDS", a.patch_bytes);
    return SM_STATUS_NOT_SUPPORTED;
  }
  static LdsOptIn lds_optin;
  if (const int rc = ensure_dyn_lds(lds_optin, reinterpret_cast<const void*>(&conv_spmma_fused_kernel<BN, BF, V16, SMALL>), 160 * 1024, "conv_spmma_fused_kernel")) return rc;
  conv_spmma_fused_kernel<BN, BF, V16, SMALL><<<dim3((unsigned)nwg), dim3(256), lds, st>>>(a);
  return check_launch("conv_spmma_fused_kernel");
}

// The geometry-only part of what the implicit kernel takes (no pointers, no n_out), and the stage plan that follows from it, for
// one DMA piece size: v16 = 16-byte pieces (8 halves per lane: W % 8 == 0; the border rounded up to 8 halves, at most 16 DMA
// instructions per stage) or 4-byte pieces (2 halves per lane, at most 48).  Shared by the launcher -- which tries the 16-byte form
// first and falls back to the 4-byte form when only the tighter limits fail -- and by sm_conv_spmma_workspace.
// Returns nullptr when the geometry fits, else what does not.
static const char* conv_geometry_plan(size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad, size_t dil, bool v16,
                                      size_t bn /*column tile: 64 or 128*/, ConvArgs& a) {
  constexpr size_t GEO_MAX = 1u << 20;
  if (kh == 0 || kw == 0 || stride == 0 || dil == 0) return "invalid argument";
  if (N > 0x7fffffffull || Cin > 0x7fffffffull || H > GEO_MAX || W > GEO_MAX || kh > GEO_MAX || kw > GEO_MAX || stride > GEO_MAX || pad > GEO_MAX ||
      dil > GEO_MAX)
    return "a geometry argument exceeds what the kernel's 32-bit indexing takes";
  const size_t sh = dil * (kh - 1) + 1, sw = dil * (kw - 1) + 1;
  if (H + 2 * pad < sh || W + 2 * pad < sw) return "window larger than the padded input";
  const size_t OH = (H + 2 * pad - sh) / stride + 1, OW = (W + 2 * pad - sw) / stride + 1, L = OH * OW, K = Cin * kh * kw;
  if (v16 && W % 8 != 0) return "16-byte patch pieces need W % 8 == 0";
  size_t padl = pad > (sw - 1 > pad ? sw - 1 - pad : 0) ? pad : (sw - 1 > pad ? sw - 1 - pad : 0);
  const size_t epl = v16 ? 8 : 2;
  padl = (padl + epl - 1) / epl * epl;
  if (padl == 0) padl = epl;  // a row's right border is the next row's left border: at least one lane's piece
  const size_t pitch = padl + W;
  // whole 64-deep stages, input rows that fit one DMA instruction with their border
  if (K == 0 || K % 64 != 0 || W % 2 != 0 || pitch / epl > 64 || kh * kw > 64 || H > 0x7fff || N * L > 0x7fffffffull || K > 0x7fffffffull)
    return "needs C*kh*kw % 64 == 0, an even W that fits one DMA instruction with its border, and kh*kw <= 64";
  a.N = (int)N; a.Cin = (int)Cin; a.H = (int)H; a.W = (int)W; a.kh = (int)kh; a.kw = (int)kw;
  a.stride = (int)stride; a.pad = (int)pad; a.dil = (int)dil; a.OH = (int)OH; a.OW = (int)OW; a.L = (int)L;
  a.K = (int)K; a.khkw = (int)(kh * kw);
  const size_t span = L < 128 ? (L - 1) / OW : (127 + OW - 1) / OW;  // most output rows a 128-pixel tile straddles, minus one
  a.RI = (int)(span * stride + sh);
  a.pitch = (int)pitch; a.padl = (int)padl;
  a.rpi = (int)(64 / (pitch / epl));
  a.nch = (int)((kh * kw - 1 + 63) / (kh * kw) + 1);
  a.a_n = (int)ceil_div((size_t)a.nch * a.RI, (size_t)a.rpi);
  a.patch_bytes = (int)round_up(((size_t)a.nch * a.RI * pitch + padl + 2 * (sw + pad)) * 2 + 256, 16);
  if (a.a_n > (v16 ? 16 : 48)) return "a stage's activation patch needs too many DMA instructions";
  // the gather's k-offset table holds 16-bit byte offsets into the patch: largest = last channel's last window element
  if (((size_t)a.nch * a.RI * pitch + kw * dil) * 2 >= 65536) return "a stage's activation patch exceeds the 16-bit offset table";
  // the kernel's LDS: two patch + B stages, the k-offset table
  if (2 * ((size_t)a.patch_bytes + 64 * bn * 2) + kh * kw * 128 > 160 * 1024) return "a stage's activation patch does not fit LDS";
  return nullptr;
}

template <bool BF>
static int conv_spmma16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw,
                        size_t stride, size_t pad, size_t dil, size_t n_out, float alpha, float beta, sm_stream_t stream) {
  const char* name = BF ? "sm_conv_spmma_fused_bf16" : "sm_conv_spmma_fused_f16";
  if (!X || !B || !C || kh == 0 || kw == 0 || stride == 0 || dil == 0) {
    set_error("%s: invalid argument", name);
    return SM_STATUS_INVALID_VALUE;
  }
  {
    const size_t sh = dil * (kh - 1) + 1, sw = dil * (kw - 1) + 1;
    if (H <= (1u << 20) && W <= (1u << 20) && pad <= (1u << 20) && kh <= (1u << 20) && kw <= (1u << 20) && dil <= (1u << 20) &&
        (H + 2 * pad < sh || W + 2 * pad < sw)) {
      set_error("%s: window larger than the padded input", name);
      return SM_STATUS_INVALID_VALUE;
    }
  }
  if (N == 0 || Cin == 0 || n_out == 0) return SM_STATUS_SUCCESS;
  // what the kernel takes beyond the geometry: n % 8 == 0, 16-byte aligned B rows, 4-byte aligned input rows; anything else:
  // sm_im2col_compress24_* + sm_spmma_* (the same result)
  if (n_out % 8 != 0 || n_out > 0x7fffffffull || !aligned16(B) || (reinterpret_cast<uintptr_t>(X) & 3u) != 0) {
    set_error("%s: needs n %% 8 == 0, a 16-byte aligned B and a 4-byte aligned X (use sm_im2col_compress24 + sm_spmma)", name);
    return SM_STATUS_NOT_SUPPORTED;
  }
  // 16-byte patch DMAs (round 4) where rows are whole 16-byte pieces: W % 8 == 0, a 16-byte aligned X, the border rounded up to
  // 8 halves and the stage's plan within 16 instructions; when only those tighter limits fail (narrow images: W = 8 gains little
  // per instruction and pays the wider border) the 4-byte form takes the layer as it did before round 4.  SM_CONV_V16 = 0 (tuning)
  // keeps the 4-byte form everywhere.
  ConvArgs a = {};
  bool v16 = W % 8 == 0 && aligned16(X) && tuning_int("SM_CONV_V16", 1) != 0;
  const size_t bn = n_out <= 64 ? 64 : 128;
  const char* why = v16 ? conv_geometry_plan(N, Cin, H, W, kh, kw, stride, pad, dil, true, bn, a) : "";
  if (why) {
    v16 = false;
    a = ConvArgs{};
    why = conv_geometry_plan(N, Cin, H, W, kh, kw, stride, pad, dil, false, bn, a);
  }
  if (why) {
    set_error("%s: %s (use sm_im2col_compress24 + sm_spmma)", name, why);
    return SM_STATUS_NOT_SUPPORTED;
  }
  a.X = (const half_t*)X; a.B = (const half_t*)B; a.C = (half_t*)C;
  a.Nout = (int)n_out;
  a.alpha = alpha; a.beta = beta;
  a.ablate = tuning_int("SM_CONV_ABLATE", 0);
  hipStream_t st = (hipStream_t)stream;
  // (256-column tiles for n_out > 128 -- the patch gathered once per 256 output channels -- were built and measured in round 4:
  //  n = 256 unchanged (73.8 vs 74.1 us), n = 512 slower (134 vs 103 us: 223 registers, two workgroups per CU, half the tiles);
  //  profiles/conv_levers_r04k.txt.  Not kept.)
  if (v16) return n_out <= 64 ? launch_conv<64, BF, true>(a, st) : launch_conv<128, BF, true>(a, st);
  if (a.a_n <= 16 && tuning_int("SM_CONV_SMALL", 1) != 0) return n_out <= 64 ? launch_conv<64, BF, false, true>(a, st) : launch_conv<128, BF, false, true>(a, st);
  if (n_out <= 64) return launch_conv<64, BF>(a, st);
  return launch_conv<128, BF>(a, st);
}

}  // namespace sm

using namespace sm;

extern "C" int sm_conv_spmma_fused_f16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh,
                                       size_t kw, size_t stride, size_t pad, size_t dilation, size_t n_out, float alpha, float beta,
                                       sm_stream_t stream) {
  return conv_spmma16<false>(X, B, C, N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, alpha, beta, stream);
}
extern "C" int sm_conv_spmma_fused_bf16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh,
                                        size_t kw, size_t stride, size_t pad, size_t dilation, size_t n_out, float alpha, float beta,
                                        sm_stream_t stream) {
  return conv_spmma16<true>(X, B, C, N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, alpha, beta, stream);
}

// ---------------------------------------------------------------------------------------------
// The convolution-layer product by the faster of its two routes (round 4): the implicit-GEMM kernel above, or -- for the small-
// spatial, long-K layers, where that kernel's per-stage gather is spread over few tiles (14 x 14 x 512 channels: 102 us against
// 45 + 37 us, profiles/conv_routes_r04ac.txt) -- sm_im2col_compress24 into the caller's workspace followed by the staged 2:4
// matmul.  Both routes give the same C bit for bit.  workspace: sm_conv_spmma_workspace bytes -- the blob's size for the layers the
// rule sends to the pair AND for every geometry the implicit kernel cannot run (K % 64 != 0 such as the 7 x 7 x 3 stem, an odd or
// too wide W, kh * kw > 64, ...: conv_geometry_plan), 0 for the layers that stay on the implicit kernel; without a workspace the
// implicit kernel runs wherever it can.
// ---------------------------------------------------------------------------------------------
static bool conv_prefers_blob(size_t out_h, size_t out_w, size_t K) { return out_h * out_w <= 256 && K >= 2048; }
static bool conv_implicit_takes_geometry(size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad, size_t dil) {
  ConvArgs a = {};
  // (the 128-column tile's LDS: n_out is not known to the query.)  The 4-byte plan only: the 16-byte plan also needs a
  // 16-byte aligned X, which the query cannot see -- a geometry ONLY that plan takes (W % 8 == 0 with W + border > 128 halves, e.g. W = 224)
  // would otherwise get "no workspace needed" and then NOT_SUPPORTED for a 4-byte aligned X with nothing to fall back on.  Such a
  // geometry is sized for the blob; sm_conv_spmma_* still runs the implicit kernel on it when X turns out 16-byte aligned.
  return conv_geometry_plan(N, Cin, H, W, kh, kw, stride, pad, dil, false, 128, a) == nullptr;
}

extern "C" int sm_conv_spmma_workspace(size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad, size_t dilation,
                                       size_t* bytes) {
  size_t oh = 0, ow = 0;
  if (!bytes || sm_conv_out_size(H, kh, stride, pad, dilation, &oh) != SM_STATUS_SUCCESS || sm_conv_out_size(W, kw, stride, pad, dilation, &ow) != SM_STATUS_SUCCESS) {
    set_error("sm_conv_spmma_workspace: invalid argument");
    return SM_STATUS_INVALID_VALUE;
  }
  *bytes = 0;
  if (N == 0 || Cin == 0) return SM_STATUS_SUCCESS;
  // 0 only where the implicit kernel both is the preferred route and can run the geometry (n_out and the operands' alignment are
  // not known here: n_out % 8 != 0 or a misaligned B / X also need the blob -- size it with sm_compress24_size then)
  if (!conv_prefers_blob(oh, ow, Cin * kh * kw) && conv_implicit_takes_geometry(N, Cin, H, W, kh, kw, stride, pad, dilation)) return SM_STATUS_SUCCESS;
  return sm_compress24_size(oh * ow, Cin * kh * kw, 2, N, bytes);
}

template <bool BF>
static int conv_spmma_routed(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride, size_t pad,
                             size_t dilation, size_t n_out, float alpha, float beta, void* workspace, size_t workspace_bytes, sm_stream_t stream) {
  size_t oh = 0, ow = 0, need = 0;
  if (sm_conv_out_size(H, kh, stride, pad, dilation, &oh) != SM_STATUS_SUCCESS || sm_conv_out_size(W, kw, stride, pad, dilation, &ow) != SM_STATUS_SUCCESS) {
    set_error("sm_conv_spmma: invalid geometry");
    return SM_STATUS_INVALID_VALUE;
  }
  const size_t L = oh * ow, K = Cin * kh * kw;
  bool blob_route = conv_prefers_blob(oh, ow, K) && workspace && sm_compress24_size(L, K, 2, N, &need) == SM_STATUS_SUCCESS && workspace_bytes >= need &&
                    aligned16(workspace);
  if (!blob_route) {
    const int rc = conv_spmma16<BF>(X, B, C, N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, alpha, beta, stream);
    if (rc != SM_STATUS_NOT_SUPPORTED) return rc;
    // a geometry the implicit kernel does not take: the pair, if the caller gave room for the blob
    if (!workspace || sm_compress24_size(L, K, 2, N, &need) != SM_STATUS_SUCCESS || workspace_bytes < need || !aligned16(workspace)) return rc;
  }
  int rc = BF ? sm_im2col_compress24_bf16(X, N, Cin, H, W, kh, kw, stride, pad, dilation, workspace, stream)
              : sm_im2col_compress24_f16(X, N, Cin, H, W, kh, kw, stride, pad, dilation, workspace, stream);
  if (rc != SM_STATUS_SUCCESS) return rc;
  return BF ? sm_spmma_bf16(workspace, B, C, L, n_out, K, N, 0, L * n_out, alpha, beta, stream) : sm_spmma_f16(workspace, B, C, L, n_out, K, N, 0, L * n_out, alpha, beta, stream);
}

extern "C" int sm_conv_spmma_f16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                                 size_t pad, size_t dilation, size_t n_out, float alpha, float beta, void* workspace, size_t workspace_bytes,
                                 sm_stream_t stream) {
  return conv_spmma_routed<false>(X, B, C, N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, alpha, beta, workspace, workspace_bytes, stream);
}
extern "C" int sm_conv_spmma_bf16(const void* X, const void* B, void* C, size_t N, size_t Cin, size_t H, size_t W, size_t kh, size_t kw, size_t stride,
                                  size_t pad, size_t dilation, size_t n_out, float alpha, float beta, void* workspace, size_t workspace_bytes,
                                  sm_stream_t stream) {
  return conv_spmma_routed<true>(X, B, C, N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, alpha, beta, workspace, workspace_bytes, stream);
}
