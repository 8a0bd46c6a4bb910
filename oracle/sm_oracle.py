"""ctypes binding of the CPU oracle (oracle/sm_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package.  Arrays are numpy; fp16 data travel as uint16 bit patterns (np.float16 views).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsm_oracle.so")
_lib = None

TILE, STRIP = 0, 1


def build():
    res = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building the oracle failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        try:
            _lib = ctypes.CDLL(LIB_PATH)
        except OSError:
            build()  # e.g. built on a different host CPU
            _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _sz(x):
    return ctypes.c_size_t(int(x))


def _ok(rc, what):
    if rc != 0:
        raise ValueError(f"oracle {what} rejected its arguments (rc={rc})")


def _sfx(a):
    return {1: "i8", 2: "f16", 4: "f32", 8: "f64"}[a.dtype.itemsize]


def sparsify_positional(weights, mask, m, n, sparsity_factor=0.5, blk_m=2, blk_n=2):
    """In place on flat `weights` (any float dtype) and `mask` (uint64)."""
    assert mask.dtype == np.uint64 and weights.flags.c_contiguous and mask.flags.c_contiguous
    _ok(lib().sm_sparsify_positional_ref(_p(weights), _p(mask), _sz(m), _sz(n), _sz(weights.dtype.itemsize),
                                         _sz(blk_m), _sz(blk_n), ctypes.c_float(sparsity_factor)), "sparsify_positional")


def prune24(A, m, k, ld, alg=STRIP, bf16=False):
    """Returns the pruned copy of row-major A (shape-agnostic flat buffer of >= (m-1)*ld + k elements).
    bf16=True: the uint16 bit patterns are bfloat16 (matters for the TILE rule only)."""
    out = A.copy()
    fn = getattr(lib(), "sm_prune24_%s_ref" % ("bf16" if bf16 else _sfx(A)))
    _ok(fn(_p(A), _p(out), _sz(m), _sz(k), _sz(ld), ctypes.c_int(alg)), "prune24")
    return out


def tile_select_both(mag):
    """(mask two-level, mask exhaustive, score two-level, score exhaustive) of one 4x4 tile of fp32 magnitudes."""
    mag = np.ascontiguousarray(mag, dtype=np.float32).reshape(16)
    ma, mb = ctypes.c_uint(0), ctypes.c_uint(0)
    sa, sb = ctypes.c_float(0), ctypes.c_float(0)
    _ok(lib().sm_tile_select_both_ref(_p(mag), ctypes.byref(ma), ctypes.byref(mb), ctypes.byref(sa), ctypes.byref(sb)), "tile_select_both")
    return ma.value, mb.value, sa.value, sb.value


def prune24_check(A, m, k, ld):
    v = ctypes.c_int(-1)
    fn = getattr(lib(), "sm_prune24_check_%s_ref" % _sfx(A))
    _ok(fn(_p(A), _sz(m), _sz(k), _sz(ld), ctypes.byref(v)), "prune24_check")
    return v.value


def compress24_size(m, k, elt_bytes, batch=1):
    out = ctypes.c_size_t(0)
    _ok(lib().sm_compress24_size_ref(_sz(m), _sz(k), _sz(elt_bytes), _sz(batch), ctypes.byref(out)), "compress24_size")
    return out.value


def compress24_layout(m, k, elt_bytes, batch=1):
    kc, mo, tot = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
    _ok(lib().sm_compress24_layout(_sz(m), _sz(k), _sz(elt_bytes), _sz(batch), ctypes.byref(kc), ctypes.byref(mo),
                                   ctypes.byref(tot)), "compress24_layout")
    return kc.value, mo.value, tot.value


def compress24(A, m, k, ld, batch=1, strideA=None):
    strideA = m * ld if strideA is None else strideA
    blob = np.zeros(compress24_size(m, k, A.dtype.itemsize, batch), dtype=np.uint8)
    fn = getattr(lib(), "sm_compress24_%s_ref" % _sfx(A))
    _ok(fn(_p(A), _sz(m), _sz(k), _sz(ld), _sz(batch), _sz(strideA), _p(blob)), "compress24")
    return blob


def decompress24(blob, m, k, ld, dtype, batch=1, strideA=None):
    strideA = m * ld if strideA is None else strideA
    A = np.zeros(batch * strideA, dtype=dtype)
    fn = getattr(lib(), "sm_decompress24_%s_ref" % _sfx(A))
    _ok(fn(_p(blob), _sz(m), _sz(k), _sz(ld), _sz(batch), _sz(strideA), _p(A)), "decompress24")
    return A


def spmma(blob, B, C, m, n, k, batch=1, strideB=0, strideC=None, alpha=1.0, beta=0.0, bf16=False):
    """C updated in place (fp64 accumulation).  bf16=True: B, C and the blob's values are bfloat16 bit patterns."""
    strideC = m * n if strideC is None else strideC
    fn = getattr(lib(), "sm_spmma_%s_ref" % ("bf16" if bf16 else _sfx(B)))
    _ok(fn(_p(blob), _p(B), _p(C), _sz(m), _sz(n), _sz(k), _sz(batch), _sz(strideB), _sz(strideC),
           ctypes.c_float(alpha), ctypes.c_float(beta)), "spmma")
    return C


def conv_out_size(size, kernel, stride=1, pad=0, dilation=1):
    out = ctypes.c_size_t(0)
    _ok(lib().sm_conv_out_size_ref(_sz(size), _sz(kernel), _sz(stride), _sz(pad), _sz(dilation), ctypes.byref(out)), "conv_out_size")
    return out.value


def im2col(X, N, C, H, W, kh, kw, stride=1, pad=0, dilation=1):
    """X: flat NCHW array (any dtype; elements move as opaque values) -> flat [N][L][C*kh*kw]."""
    OH, OW = conv_out_size(H, kh, stride, pad, dilation), conv_out_size(W, kw, stride, pad, dilation)
    A = np.zeros(N * OH * OW * C * kh * kw, dtype=X.dtype)
    _ok(lib().sm_im2col_ref(_p(X), _sz(N), _sz(C), _sz(H), _sz(W), _sz(kh), _sz(kw), _sz(stride), _sz(pad), _sz(dilation),
                            _sz(X.dtype.itemsize), _p(A)), "im2col")
    return A


def gemm_batched(As, Bs, Cs, m, n, k, alpha=1.0, beta=0.0, ta=0, tb=0):
    """Column-major pointer-array GEMM (gemm.hxx:80-81); As/Bs/Cs are lists of flat arrays; Cs in place."""
    batch = len(Cs)
    arr = lambda xs: (ctypes.c_void_p * batch)(*[x.ctypes.data for x in xs])
    sfx = _sfx(Cs[0])
    fn = getattr(lib(), "sm_gemm_batched_%s_ref" % sfx)
    sc = ctypes.c_double if sfx == "f64" else ctypes.c_float
    _ok(fn(arr(As), arr(Bs), arr(Cs), _sz(m), _sz(n), _sz(k), _sz(batch), ctypes.c_int(ta), ctypes.c_int(tb),
           sc(alpha), sc(beta)), "gemm_batched")
    return Cs


def spmma_i8(blob, B, C, m, n, k, batch=1, strideB=0, strideC=None, accumulate=False):
    """int8 2:4 product: B is [n][k] (k-contiguous per output column) int8, C int32 in place; exact."""
    strideC = m * n if strideC is None else strideC
    assert B.dtype == np.int8 and C.dtype == np.int32
    _ok(lib().sm_spmma_i8_ref(_p(blob), _p(B), _p(C), _sz(m), _sz(n), _sz(k), _sz(batch), _sz(strideB), _sz(strideC),
                              ctypes.c_int(1 if accumulate else 0)), "spmma_i8")
    return C


def requant_i8(acc, scale):
    out = np.zeros(acc.size, dtype=np.int8)
    _ok(lib().sm_requant_i8_ref(_p(acc), _p(out), _sz(acc.size), ctypes.c_float(scale)), "requant_i8")
    return out


def gemm_rowmajor(A, B, C, m, n, k, lda=None, batch=1, strideA=None, strideB=0, strideC=None, alpha=1.0, beta=0.0, bf16=False):
    lda = k if lda is None else lda
    strideA = m * lda if strideA is None else strideA
    strideC = m * n if strideC is None else strideC
    fn = getattr(lib(), "sm_gemm_rowmajor_%s_ref" % ("bf16" if bf16 else _sfx(A)))
    _ok(fn(_p(A), _p(B), _p(C), _sz(m), _sz(n), _sz(k), _sz(lda), _sz(batch), _sz(strideA), _sz(strideB),
           _sz(strideC), ctypes.c_float(alpha), ctypes.c_float(beta)), "gemm_rowmajor")
    return C


def spmm_bell(values, column_indices, rows, cols, block_size, ell_cols, B, C, n, alpha=1.0, beta=0.0):
    _ok(lib().sm_spmm_bell_f32_ref(_p(values), _p(column_indices), _sz(rows), _sz(cols), _sz(block_size),
                                   _sz(ell_cols), _p(B), _p(C), _sz(n), ctypes.c_float(alpha), ctypes.c_float(beta)),
        "spmm_bell")
    return C


def spmm_coo(A_rows, A_cols, nnz, B_cols, batches, rows, cols, vals, B, C, alpha=1.0, beta=0.0):
    _ok(lib().sm_spmm_coo_f32_ref(_sz(A_rows), _sz(A_cols), _sz(nnz), _sz(B_cols), _sz(batches), _p(rows), _p(cols),
                                  _p(vals), _p(B), _p(C), ctypes.c_float(alpha), ctypes.c_float(beta)), "spmm_coo")
    return C


# ---- timed CPU baseline (bench.py cpu_baseline leg only) ----
def cpu_gemm_f32(A, B, C, m, n, k):
    lib().sm_cpu_gemm_f32(_p(A), _p(B), _p(C), _sz(m), _sz(n), _sz(k))


def cpu_spmma_f32(A, B, C, m, n, k):
    lib().sm_cpu_spmma_f32(_p(A), _p(B), _p(C), _sz(m), _sz(n), _sz(k))


def num_threads():
    return int(lib().sm_oracle_num_threads())
