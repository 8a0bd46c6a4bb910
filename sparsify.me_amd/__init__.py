"""sparsify.me_amd -- ctypes front end of libsparsifyme.so (the gfx950 HIP back end).

This package is plumbing for tests and bench.py: it hands torch-owned device pointers to the
C ABI declared in include/sparsifyme.h.  There is NO CPU fallback here: if the HIP library is
missing or a call fails, the functions raise.  The C++ mirror of the reference's operator API
lives in include/sparsify.me/*.hxx; the function names below follow it
(sparsify, spmma, batched gemm -- reference include/sparsify.me/{sparsify,spmma,gemm}.hxx).

The directory name contains a dot, so import it through __graft_entry__.load_package(), which
registers it as module `sparsifyme_amd`.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPARSIFYME_LIB: diagnostic builds only (e.g. the cycle-stamped library of tools/); default is the in-tree product library
LIB_PATH = os.environ.get("SPARSIFYME_LIB") or os.path.join(_HERE, "libsparsifyme.so")

PRUNE_TILE = 0
PRUNE_STRIP = 1
# include/sparsifyme.h: SM_STATUS_*
STATUS_SUCCESS, STATUS_INVALID_VALUE, STATUS_NOT_SUPPORTED, STATUS_LAUNCH_FAILED, STATUS_NO_DEVICE = 0, 1, 2, 3, 4

_lib = None

_c_size = ctypes.c_size_t
_c_ptr = ctypes.c_void_p
_c_f = ctypes.c_float
_c_i = ctypes.c_int

# name -> argtypes (restype is always int unless listed in _RET)
_SIGS = {
    "sm_device_check": [],
    "sm_sparsify_positional": [_c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_f, _c_ptr],
    "sm_sparsify_positional_f16": [_c_ptr, _c_ptr, _c_size, _c_size, _c_f, _c_ptr],
    "sm_sparsify_positional_f32": [_c_ptr, _c_ptr, _c_size, _c_size, _c_f, _c_ptr],
    "sm_sparsify_positional_f64": [_c_ptr, _c_ptr, _c_size, _c_size, _c_f, _c_ptr],
    "sm_prune24_f16": [_c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_i, _c_ptr],
    "sm_prune24_f32": [_c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_i, _c_ptr],
    "sm_prune24_check_f16": [_c_ptr, _c_size, _c_size, _c_size, _c_ptr, _c_ptr],
    "sm_prune24_check_f32": [_c_ptr, _c_size, _c_size, _c_size, _c_ptr, _c_ptr],
    "sm_compress24_size": [_c_size, _c_size, _c_size, _c_size, ctypes.POINTER(_c_size)],
    "sm_compress24_f16": [_c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr],
    "sm_compress24_f32": [_c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr],
    "sm_decompress24_f16": [_c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr],
    "sm_decompress24_f32": [_c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr],
    "sm_spmma_f16": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmma_f32": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmma_fused_f16": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size,
                           _c_f, _c_f, _c_ptr],
    "sm_spmma_fused_f32": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size,
                           _c_f, _c_f, _c_ptr],
    "sm_conv_spmma_workspace": [_c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, ctypes.POINTER(_c_size)],
    "sm_conv_spmma_f16": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_f, _c_f,
                          _c_ptr, _c_size, _c_ptr],
    "sm_conv_spmma_bf16": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_f, _c_f,
                           _c_ptr, _c_size, _c_ptr],
    "sm_gemm_rowmajor_f32_split": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_i,
                                   _c_ptr, _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmma_f16_grouped": [_c_size, _c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmma_bf16_grouped": [_c_size, _c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmma_fused_f32_split_workspace": [_c_size, _c_size, _c_size, _c_size, _c_i, ctypes.POINTER(_c_size)],
    "sm_spmma_fused_f32_split": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_i,
                                 _c_ptr, _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmma_fused_f16_grouped": [_c_size, _c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size,
                                   _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmma_fused_bf16_grouped": [_c_size, _c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size,
                                    _c_size, _c_f, _c_f, _c_ptr],
    "sm_gemm_batched_f16": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_i, _c_i, _c_f, _c_f, _c_ptr],
    "sm_gemm_batched_f32": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_i, _c_i, _c_f, _c_f, _c_ptr],
    "sm_gemm_batched_f64": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_i, _c_i,
                            ctypes.c_double, ctypes.c_double, _c_ptr],
    "sm_gemm_rowmajor_f16": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size,
                             _c_size, _c_f, _c_f, _c_ptr],
    "sm_gemm_rowmajor_f32": [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size,
                             _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmm_bell_f32": [_c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr, _c_size, _c_f, _c_f, _c_ptr],
    "sm_spmm_coo_f32": [_c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f, _c_f,
                        _c_ptr],
    "sm_spmm_bell_workspace_size": [_c_size, _c_size, ctypes.POINTER(_c_size)],
    "sm_spmm_bell_f32_ws": [_c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr, _c_size, _c_f, _c_f, _c_ptr,
                            _c_ptr],
    "sm_spmm_bell_batched_workspace_size": [_c_size, _c_size, _c_size, ctypes.POINTER(_c_size)],
    "sm_spmm_bell_batched_f32": [_c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr, _c_size, _c_size, _c_f,
                                 _c_f, _c_ptr, _c_ptr],
    "sm_spmm_coo_workspace_size": [_c_size, ctypes.POINTER(_c_size)],
    "sm_spmm_coo_packed_workspace_size": [_c_size, _c_size, ctypes.POINTER(_c_size)],
    "sm_spmm_coo_f32_packed": [_c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f, _c_f,
                               _c_ptr, _c_size, _c_ptr],
    "sm_spmm_coo_f32_ws": [_c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f, _c_f,
                           _c_ptr, _c_ptr],
    "sm_spmm_coo_fast_workspace_size": [_c_size, _c_size, _c_size, _c_size, ctypes.POINTER(_c_size)],
    "sm_spmm_coo_fast_flag": [_c_ptr, ctypes.POINTER(_c_i), _c_ptr],
    "sm_spmm_coo_fast_form": [_c_size, _c_size, _c_size, _c_size, _c_size, _c_f],
    "sm_spmm_coo_f32_fast": [_c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_f, _c_f,
                             _c_ptr, _c_size, _c_ptr],
    "sm_fill_uniform_f16": [_c_ptr, _c_size, ctypes.c_uint64, _c_f, _c_f, _c_ptr],
    "sm_fill_uniform_f32": [_c_ptr, _c_size, ctypes.c_uint64, _c_f, _c_f, _c_ptr],
    "sm_copy_bytes": [_c_ptr, _c_ptr, _c_size, _c_ptr],
}
_SIGS["sm_prune24_compress24_f16"] = [_c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_ptr, _c_ptr, _c_i, _c_ptr]
_SIGS["sm_prune24_compress24_bf16"] = _SIGS["sm_prune24_compress24_f16"]
_SIGS["sm_prune24_spmma_f16"] = [_c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_size] * 8 + [_c_i, _c_ptr, _c_f, _c_f, _c_ptr]
_SIGS["sm_prune24_spmma_bf16"] = _SIGS["sm_prune24_spmma_f16"]
_SIGS["sm_prune24_compress24_f32"] = _SIGS["sm_prune24_compress24_f16"]
_SIGS["sm_conv_spmma_fused_f16"] = [_c_ptr, _c_ptr, _c_ptr] + [_c_size] * 10 + [_c_f, _c_f, _c_ptr]
_SIGS["sm_conv_spmma_fused_bf16"] = _SIGS["sm_conv_spmma_fused_f16"]
_SIGS["sm_transpose"] = [_c_ptr, _c_ptr] + [_c_size] * 8 + [_c_ptr]
_SIGS["sm_conv_out_size"] = [_c_size, _c_size, _c_size, _c_size, _c_size, ctypes.POINTER(_c_size)]
_SIGS["sm_im2col_f16"] = [_c_ptr] + [_c_size] * 9 + [_c_ptr, _c_ptr]
_SIGS["sm_im2col_compress24_f16"] = [_c_ptr] + [_c_size] * 9 + [_c_ptr, _c_ptr]
for _name in ("sm_prune24", "sm_prune24_check", "sm_compress24", "sm_decompress24"):
    _SIGS[_name + "_i8"] = _SIGS[_name + "_f16"]
_SIGS["sm_spmma_fused_i8"] = [_c_ptr, _c_ptr, _c_ptr] + [_c_size] * 8 + [_c_i, _c_ptr]
_SIGS["sm_spmma_fused_i8_q"] = [_c_ptr, _c_ptr, _c_ptr] + [_c_size] * 8 + [_c_f, _c_ptr]
_SIGS["sm_transpose_i8"] = [_c_ptr, _c_ptr, _c_size, _c_size, _c_ptr]
_SIGS["sm_spmma_i8"] = [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_i, _c_ptr]
_SIGS["sm_spmma_i8_q"] = [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_size, _c_size, _c_f, _c_ptr]
# bfloat16 forms: same signatures as their _f16 counterparts
for _name in ("sm_prune24", "sm_prune24_check", "sm_compress24", "sm_decompress24", "sm_spmma", "sm_spmma_fused",
              "sm_gemm_rowmajor", "sm_fill_uniform", "sm_im2col", "sm_im2col_compress24"):
    _SIGS[_name + "_bf16"] = _SIGS[_name + "_f16"]
_SIGS["sm_spmma_fused_f32_split_prepare"] = [_c_ptr, _c_size, _c_size, _c_size, _c_size, _c_i, _c_ptr, _c_size, _c_ptr]
_SIGS["sm_spmma_fused_f32_split_prepared"] = [_c_ptr, _c_ptr, _c_ptr] + [_c_size] * 8 + [_c_i, _c_size, _c_f, _c_f, _c_ptr]
_SIGS["sm_spmma_fused_workspace_size"] = [ctypes.POINTER(_c_size)]
_SIGS["sm_spmma_fused_workspace_state"] = [_c_ptr, ctypes.POINTER(_c_i), _c_ptr]
_SIGS["sm_spmma_fused_streamk_plan"] = [_c_size, _c_size, _c_size, _c_size, ctypes.POINTER(_c_i), ctypes.POINTER(ctypes.c_uint)]
_SIGS["sm_gemm_rowmajor_f16_ws"] = [_c_ptr, _c_ptr, _c_ptr] + [_c_size] * 8 + [_c_f, _c_f, _c_ptr, _c_size, _c_ptr]
_SIGS["sm_gemm_rowmajor_bf16_ws"] = _SIGS["sm_gemm_rowmajor_f16_ws"]
_SIGS["sm_gemm_batched_f16_ws"] = [_c_ptr, _c_ptr, _c_ptr, _c_size, _c_size, _c_size, _c_size, _c_i, _c_i, _c_f, _c_f, _c_ptr, _c_size, _c_ptr]
_SIGS["sm_spmma_fused_f16_ws"] = [_c_ptr, _c_ptr, _c_ptr] + [_c_size] * 8 + [_c_f, _c_f, _c_ptr, _c_size, _c_ptr]
_SIGS["sm_spmma_fused_bf16_ws"] = _SIGS["sm_spmma_fused_f16_ws"]
_SIGS["sm_spmma_fused_f16_grouped_ws"] = [_c_size, _c_ptr, _c_ptr, _c_ptr] + [_c_size] * 8 + [_c_f, _c_f, _c_ptr, _c_size, _c_ptr]
_SIGS["sm_spmma_fused_bf16_grouped_ws"] = _SIGS["sm_spmma_fused_f16_grouped_ws"]
_RET = {"sm_version": ctypes.c_char_p, "sm_last_error": ctypes.c_char_p}

# every symbol include/sparsifyme.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = sorted(list(_SIGS) + list(_RET))


class SparsifymeError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libsparsifyme.so (hipcc cross-compiles without a GPU)."""
    res = subprocess.run(["make", "-C", _HERE, "-j4"], capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise SparsifymeError("building libsparsifyme.so failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return LIB_PATH


def lib():
    """The loaded HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SparsifymeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the product path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        for name, ret in _RET.items():
            fn = getattr(L, name)
            fn.argtypes = []
            fn.restype = ret
        _lib = L
    return _lib


def _check(status, what):
    if status != 0:
        msg = lib().sm_last_error().decode()
        raise SparsifymeError(f"{what} failed with status {status}: {msg}")


_torch = None
_SFX = None


def _t():
    global _torch, _SFX
    if _torch is None:
        import torch
        _torch = torch
        _SFX = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32", torch.float64: "f64", torch.int8: "i8"}
    return _torch


def _stream():
    return ctypes.c_void_p(_t().cuda.current_stream().cuda_stream)


def _dev(t):
    if not t.is_cuda:
        raise SparsifymeError("expected a device tensor (the product path has no host implementation)")
    return ctypes.c_void_p(t.data_ptr())


def _sfx(t):
    _t()
    return _SFX[t.dtype]


def graph_time_ms(fn, iters=20, replays=3):
    """Device time of one call of `fn` in ms: `iters` calls are captured into one hipGraph (so the
    host-side ctypes/launch cost is not on the clock) and the graph is replayed `replays` times
    between two HIP events on the capture stream's parent.  fn must only enqueue work on the
    current stream."""
    torch = _t()
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * replays)


def version():
    return lib().sm_version().decode()


def device_check():
    _check(lib().sm_device_check(), "sm_device_check")


# ---- operators -------------------------------------------------------------------------------
def sparsify(weights, mask, m, n, sparsity_factor=0.5, blk_m=2, blk_n=2):
    """sparsifyme::sparsify<BLK_M,BLK_N> (sparsify.hxx:24-30): in place on `weights`
    (m*n elements) and `mask` (m*n int64/uint64)."""
    assert weights.numel() >= m * n and mask.numel() >= m * n and mask.element_size() == 8
    _check(lib().sm_sparsify_positional(_dev(weights), _dev(mask), m, n, weights.element_size(), blk_m, blk_n,
                                        float(sparsity_factor), _stream()), "sm_sparsify_positional")


def prune24(A_in, A_out, m, k, ld, alg=PRUNE_STRIP):
    fn = getattr(lib(), "sm_prune24_" + _sfx(A_in))
    _check(fn(_dev(A_in), _dev(A_out), m, k, ld, alg, _stream()), "sm_prune24")


def prune24_check(A, m, k, ld, d_valid):
    fn = getattr(lib(), "sm_prune24_check_" + _sfx(A))
    _check(fn(_dev(A), m, k, ld, _dev(d_valid), _stream()), "sm_prune24_check")


def compress24_size(m, k, elt_bytes, batch=1):
    out = ctypes.c_size_t(0)
    _check(lib().sm_compress24_size(m, k, elt_bytes, batch, ctypes.byref(out)), "sm_compress24_size")
    return out.value


def compress24(A, m, k, ld, batch, strideA, blob):
    fn = getattr(lib(), "sm_compress24_" + _sfx(A))
    _check(fn(_dev(A), m, k, ld, batch, strideA, _dev(blob), _stream()), "sm_compress24")


def prune24_compress24(A_in, A_out, m, k, ld, batch, strideA, blob, d_valid, alg=PRUNE_TILE):
    """prune -> check -> compress in one pass (spmma.hxx:82-104); A_out / blob / d_valid may be None."""
    fn = getattr(lib(), "sm_prune24_compress24_" + _sfx(A_in))
    opt = lambda t: _dev(t) if t is not None else None
    _check(fn(_dev(A_in), opt(A_out), m, k, ld, batch, strideA, opt(blob), opt(d_valid), alg, _stream()), "sm_prune24_compress24")


API_SPMMA_SEQUENCE = ("sm_prune24_compress24 (TILE prune out of place from the dense A + check flag + blob, one pass) -> "
                      "sm_spmma; the flag stays on the device (the reference reads it back and synchronises, spmma.hxx:89-92)")


def api_spmma_step(A, Apruned, B, C, blob, d_valid, m, n, k, batch):
    """One call of sparsifyme::spmma() as device work (reference spmma.hxx:82-113): TILE prune (A -> Apruned, the bytes of
    an in-place prune without destroying the bench's operand), check, compress, multiply."""
    prune24_compress24(A, Apruned, m, k, k, batch, m * k, blob, d_valid, PRUNE_TILE)
    spmma(blob, B, C, m, n, k, batch, 0)


def prune24_spmma(A_in, A_out, B, C, m, n, k, lda=None, batch=1, strideA=None, strideB=0, strideC=None, alg=PRUNE_TILE, d_valid=None,
                  alpha=1.0, beta=0.0, check=True):
    """sparsifyme::spmma()'s whole sequence in one kernel (spmma.hxx:82-113): prune A_in -> A_out (may be A_in), flag, multiply.
    Returns the status (check=False: SM_STATUS_NOT_SUPPORTED is returned, not raised, so that callers can fall back)."""
    lda = k if lda is None else lda
    strideA = m * lda if strideA is None else strideA
    strideC = m * n if strideC is None else strideC
    fn = getattr(lib(), "sm_prune24_spmma_" + _sfx(A_in))
    rc = fn(_dev(A_in), _dev(A_out), _dev(B), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC, alg,
            _dev(d_valid) if d_valid is not None else None, alpha, beta, _stream())
    if check or rc not in (0, STATUS_NOT_SUPPORTED):
        _check(rc, "sm_prune24_spmma")
    return rc


def api_spmma_step_fused(A, Apruned, B, C, blob, d_valid, m, n, k, batch):
    """api_spmma_step without a blob wherever sm_prune24_spmma_* takes the shape (one kernel for n <= 128, k % 64 == 0, m % 4 == 0; since round 6
    the prune + flag pass followed by the exact fused kernel on the pruned operand elsewhere), the blob pair where it does not."""
    if prune24_spmma(A, Apruned, B, C, m, n, k, batch=batch, d_valid=d_valid, check=False) == STATUS_NOT_SUPPORTED:
        api_spmma_step(A, Apruned, B, C, blob, d_valid, m, n, k, batch)


def decompress24(blob, m, k, ld, batch, strideA, A):
    fn = getattr(lib(), "sm_decompress24_" + _sfx(A))
    _check(fn(_dev(blob), m, k, ld, batch, strideA, _dev(A), _stream()), "sm_decompress24")


def spmma(blob, B, C, m, n, k, batch=1, strideB=0, strideC=None, alpha=1.0, beta=0.0):
    """The matmul step of sparsifyme::spmma (spmma.hxx:112-113) on a compressed blob."""
    if strideC is None:
        strideC = m * n
    fn = getattr(lib(), "sm_spmma_" + _sfx(B))
    _check(fn(_dev(blob), _dev(B), _dev(C), m, n, k, batch, strideB, strideC, float(alpha), float(beta), _stream()),
           "sm_spmma")


def spmma_fused_i8(A, B, C, m, n, k, lda=None, batch=1, strideA=None, strideB=0, strideC=None, accumulate=False, scale=None):
    """int8 prune + compress + matmul in one kernel; C int32 (scale None) or int8 requantised with `scale`."""
    lda = k if lda is None else lda
    strideA = m * lda if strideA is None else strideA
    strideC = m * n if strideC is None else strideC
    if scale is None:
        _check(lib().sm_spmma_fused_i8(_dev(A), _dev(B), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC,
                                       1 if accumulate else 0, _stream()), "sm_spmma_fused_i8")
    else:
        _check(lib().sm_spmma_fused_i8_q(_dev(A), _dev(B), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC, float(scale),
                                         _stream()), "sm_spmma_fused_i8_q")


def transpose_i8(src, dst, rows, cols):
    _check(lib().sm_transpose_i8(_dev(src), _dev(dst), rows, cols, _stream()), "sm_transpose_i8")


def spmma_i8(blob, B, C, m, n, k, batch=1, strideB=0, strideC=None, accumulate=False):
    """int8 2:4 product: B [n][k] int8 (k-contiguous per output column), C int32."""
    strideC = m * n if strideC is None else strideC
    if C.dtype == _t().int8:
        raise SparsifymeError("spmma_i8 writes int32; use spmma_i8_q for a requantised int8 result")
    _check(lib().sm_spmma_i8(_dev(blob), _dev(B), _dev(C), m, n, k, batch, strideB, strideC, 1 if accumulate else 0, _stream()),
           "sm_spmma_i8")


def spmma_i8_q(blob, B, C, m, n, k, scale, batch=1, strideB=0, strideC=None):
    """int8 2:4 product requantised to int8: C = saturate(rne(scale * acc))."""
    strideC = m * n if strideC is None else strideC
    _check(lib().sm_spmma_i8_q(_dev(blob), _dev(B), _dev(C), m, n, k, batch, strideB, strideC, float(scale), _stream()),
           "sm_spmma_i8_q")


def spmma_fused_workspace_size():
    out = _c_size(0)
    _check(lib().sm_spmma_fused_workspace_size(ctypes.byref(out)), "sm_spmma_fused_workspace_size")
    return out.value


def spmma_fused_streamk_plan(rows, n, k, problems=1):
    """(takes, plan): whether the workspace entry points run the stream-K form on `problems` problems of rows x n x k, and its
    decomposition as a dict (tg, wg, groups_full, tgl, wgl, slots, units, cut, cutl) -- sm_spmma_fused_streamk_plan."""
    takes = _c_i(0)
    buf = (ctypes.c_uint * 32)()   # the entry point writes plan[0 .. 24] (include/sparsifyme.h: >= 25 unsigned)
    _check(lib().sm_spmma_fused_streamk_plan(rows, n, k, problems, ctypes.byref(takes), buf), "sm_spmma_fused_streamk_plan")
    v = list(buf)
    return bool(takes.value), dict(tg=v[0], wg=v[1], groups_full=v[2], tgl=v[3], wgl=v[4], slots=v[5], units=v[6], cut=v[7:16], cutl=v[16:25])


def streamk_whole_panels(plan, panels, nkt):
    """The row panels (0 .. panels - 1) that lie inside ONE slot range of `plan` (their tiles equal the no-workspace result bit for bit)."""
    cuts = set()
    for g in range(plan["groups_full"]):
        cuts.update(g * plan["tg"] * nkt + c for c in plan["cut"][:plan["wg"] + 1])
    base = plan["groups_full"] * plan["tg"] * nkt
    cuts.update(base + c for c in plan["cutl"][:plan["wgl"] + 1])
    return [t for t in range(panels) if not any(t * nkt < u < (t + 1) * nkt for u in cuts)]


def spmma_fused_workspace_state(ws):
    """0 = the workspace's flag page is clean; 1 = a stream-K fix-up timed out (that launch's C is invalid); 2 = flags raised without a recorded
    timeout.  Blocks on the current stream (sm_spmma_fused_workspace_state); after 1 / 2 zero the first 4096 bytes before the next call."""
    st = _c_i(-1)
    _check(lib().sm_spmma_fused_workspace_state(_dev(ws), ctypes.byref(st), _stream()), "sm_spmma_fused_workspace_state")
    return st.value


def spmma_fused_workspace():
    """A zeroed workspace for the `workspace=` argument of spmma_fused / spmma_fused_grouped (the stream-K form; one per concurrent call)."""
    return _t().zeros(spmma_fused_workspace_size(), dtype=_t().uint8, device="cuda")


def spmma_fused(A, B, C, m, n, k, lda=None, batch=1, strideA=None, strideB=0, strideC=None, alpha=1.0, beta=0.0, workspace=None):
    """Fused prune(STRIP) + compress + 2:4 matmul straight from the dense A (no blob).  workspace (spmma_fused_workspace()): the
    library may run the stream-K form where whole-tile rounds leave CUs idle."""
    lda = k if lda is None else lda
    strideA = m * lda if strideA is None else strideA
    strideC = m * n if strideC is None else strideC
    if workspace is not None:
        fn = getattr(lib(), "sm_spmma_fused_%s_ws" % _sfx(A))
        _check(fn(_dev(A), _dev(B), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC, float(alpha), float(beta), _dev(workspace),
                  workspace.numel() * workspace.element_size(), _stream()), "sm_spmma_fused_ws")
        return
    fn = getattr(lib(), "sm_spmma_fused_" + _sfx(A))
    _check(fn(_dev(A), _dev(B), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC, float(alpha), float(beta), _stream()),
           "sm_spmma_fused")


def spmma_grouped(blobs, Bs, Cs, m, n, k, batch=1, strideB=0, strideC=None, alpha=1.0, beta=0.0):
    """The matmul step on len(blobs) same-shape compressed operands in one grid per 8 (sm_spmma_*_grouped): the same C as that many
    spmma() calls."""
    if not (len(blobs) == len(Bs) == len(Cs)):
        raise SparsifymeError("spmma_grouped: operand lists differ in length")
    if not blobs:
        return
    if strideC is None:
        strideC = m * n
    fn = getattr(lib(), "sm_spmma_%s_grouped" % _sfx(Bs[0]))
    _check(fn(len(blobs), _ptr_table(blobs), _ptr_table(Bs), _ptr_table(Cs), m, n, k, batch, strideB, strideC, float(alpha), float(beta),
              _stream()), "sm_spmma_grouped")


def spmma_fused_f32_split_workspace(n, k, batch=1, strideB=0, planes=3):
    out = _c_size(0)
    _check(lib().sm_spmma_fused_f32_split_workspace(n, k, batch, strideB, planes, ctypes.byref(out)), "sm_spmma_fused_f32_split_workspace")
    return out.value


def spmma_fused_f32_split(A, B, C, m, n, k, workspace, lda=None, batch=1, strideA=None, strideB=0, strideC=None, planes=3, alpha=1.0,
                          beta=0.0, check=True, dense=False):
    """fp32 2:4 product on the sparse matrix instruction through exact bfloat16 splits (sm_spmma_fused_f32_split); `workspace`:
    a uint8 tensor of spmma_fused_f32_split_workspace(...) bytes.  Returns the status (check=False: NOT_SUPPORTED is returned)."""
    lda = k if lda is None else lda
    strideA = m * lda if strideA is None else strideA
    strideC = m * n if strideC is None else strideC
    fn = lib().sm_gemm_rowmajor_f32_split if dense else lib().sm_spmma_fused_f32_split  # dense: every element of A multiplied
    rc = fn(_dev(A), _dev(B), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC, planes, _dev(workspace),
            workspace.numel() * workspace.element_size(), float(alpha), float(beta), _stream())
    if check or rc not in (0, STATUS_NOT_SUPPORTED):
        _check(rc, "sm_spmma_fused_f32_split")
    return rc


def spmma_fused_f32_split_prepare(B, n, k, workspace, batch=1, strideB=0, planes=3):
    """Split B (fp32, k x n) into its bfloat16 planes once (weights that stay the same across calls): sm_spmma_fused_f32_split_prepare."""
    _check(lib().sm_spmma_fused_f32_split_prepare(_dev(B), n, k, batch, strideB, planes, _dev(workspace), workspace.numel() * workspace.element_size(), _stream()),
           "sm_spmma_fused_f32_split_prepare")


def spmma_fused_f32_split_prepared(A, workspace, C, m, n, k, lda=None, batch=1, strideA=None, strideB=0, strideC=None, planes=3, alpha=1.0, beta=0.0, check=True):
    """The split-form product from prepared planes of B (spmma_fused_f32_split_prepare): same C as spmma_fused_f32_split, no per-call pass over B."""
    lda = k if lda is None else lda
    strideA = m * lda if strideA is None else strideA
    strideC = m * n if strideC is None else strideC
    rc = lib().sm_spmma_fused_f32_split_prepared(_dev(A), _dev(workspace), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC, planes,
                                                 workspace.numel() * workspace.element_size(), float(alpha), float(beta), _stream())
    if check or rc not in (0, STATUS_NOT_SUPPORTED):
        _check(rc, "sm_spmma_fused_f32_split_prepared")
    return rc


def _ptr_table(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[_dev(t).value for t in tensors])
    return arr


def spmma_fused_grouped(As, Bs, Cs, m, n, k, lda=None, batch=1, strideA=None, strideB=0, strideC=None, alpha=1.0, beta=0.0, workspace=None):
    """len(As) same-shape problems in one grid per 8 (sm_spmma_fused_*_grouped): the same C as len(As) spmma_fused calls."""
    if not (len(As) == len(Bs) == len(Cs)):
        raise SparsifymeError("spmma_fused_grouped: operand lists differ in length")
    if not As:
        return
    lda = k if lda is None else lda
    strideA = m * lda if strideA is None else strideA
    strideC = m * n if strideC is None else strideC
    if workspace is not None:
        fn = getattr(lib(), "sm_spmma_fused_%s_grouped_ws" % _sfx(As[0]))
        _check(fn(len(As), _ptr_table(As), _ptr_table(Bs), _ptr_table(Cs), m, n, k, lda, batch, strideA, strideB, strideC,
                  float(alpha), float(beta), _dev(workspace), workspace.numel() * workspace.element_size(), _stream()), "sm_spmma_fused_grouped_ws")
        return
    fn = getattr(lib(), "sm_spmma_fused_%s_grouped" % _sfx(As[0]))
    _check(fn(len(As), _ptr_table(As), _ptr_table(Bs), _ptr_table(Cs), m, n, k, lda, batch, strideA, strideB, strideC,
              float(alpha), float(beta), _stream()), "sm_spmma_fused_grouped")


def transpose(src, dst, rows, cols, ld_in=None, ld_out=None, batch=1, stride_in=None, stride_out=None):
    """dst[b][c][r] = src[b][r][c] (row-major, out of place): the transposed operands of sparsifyme::spmma."""
    ld_in = cols if ld_in is None else ld_in
    ld_out = rows if ld_out is None else ld_out
    stride_in = rows * ld_in if stride_in is None else stride_in
    stride_out = cols * ld_out if stride_out is None else stride_out
    _check(lib().sm_transpose(_dev(src), _dev(dst), rows, cols, ld_in, ld_out, src.element_size(), batch, stride_in, stride_out,
                              _stream()), "sm_transpose")


def conv_out_size(size, kernel, stride=1, pad=0, dilation=1):
    out = ctypes.c_size_t(0)
    _check(lib().sm_conv_out_size(size, kernel, stride, pad, dilation, ctypes.byref(out)), "sm_conv_out_size")
    return out.value


def im2col(X, N, C, H, W, kh, kw, stride, pad, dilation, out, compress=False):
    """NCHW activations -> the matmul operand A [N][L][C*kh*kw] (compress=False) or its 2:4 blob (compress=True)."""
    fn = getattr(lib(), ("sm_im2col_compress24_" if compress else "sm_im2col_") + _sfx(X))
    _check(fn(_dev(X), N, C, H, W, kh, kw, stride, pad, dilation, _dev(out), _stream()), "sm_im2col")


def conv_spmma_fused(X, B, C, N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, alpha=1.0, beta=0.0):
    """Implicit-GEMM 2:4 convolution matmul straight from NCHW activations (no dense A, no blob)."""
    fn = getattr(lib(), "sm_conv_spmma_fused_" + _sfx(X))
    _check(fn(_dev(X), _dev(B), _dev(C), N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, float(alpha), float(beta), _stream()),
           "sm_conv_spmma_fused")


def conv_spmma_workspace(N, Cin, H, W, kh, kw, stride, pad, dilation):
    out = _c_size(0)
    _check(lib().sm_conv_spmma_workspace(N, Cin, H, W, kh, kw, stride, pad, dilation, ctypes.byref(out)), "sm_conv_spmma_workspace")
    return out.value


def conv_spmma(X, B, C, N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, workspace=None, alpha=1.0, beta=0.0):
    """The convolution-layer 2:4 product by the faster route (implicit GEMM, or im2col-to-blob + matmul for small-spatial long-K layers)."""
    fn = getattr(lib(), "sm_conv_spmma_" + _sfx(X))
    wb = workspace.numel() * workspace.element_size() if workspace is not None else 0
    _check(fn(_dev(X), _dev(B), _dev(C), N, Cin, H, W, kh, kw, stride, pad, dilation, n_out, float(alpha), float(beta),
              _dev(workspace) if workspace is not None else None, wb, _stream()), "sm_conv_spmma")


def gemm_batched(A_ptrs, B_ptrs, C_ptrs, m, n, k, batch, dtype_suffix, alpha=1.0, beta=0.0, ta=0, tb=0, workspace=None):
    """sparsifyme::batched::gemm (gemm.hxx:25-36): column-major, device pointer arrays (int64 tensors)."""
    if workspace is not None and dtype_suffix == "f16":
        _check(lib().sm_gemm_batched_f16_ws(_dev(A_ptrs), _dev(B_ptrs), _dev(C_ptrs), m, n, k, batch, ta, tb, alpha, beta, _dev(workspace),
                                            workspace.numel() * workspace.element_size(), _stream()), "sm_gemm_batched_ws")
        return
    fn = getattr(lib(), "sm_gemm_batched_" + dtype_suffix)
    _check(fn(_dev(A_ptrs), _dev(B_ptrs), _dev(C_ptrs), m, n, k, batch, ta, tb, alpha, beta, _stream()),
           "sm_gemm_batched")


def gemm_rowmajor(A, B, C, m, n, k, lda=None, batch=1, strideA=None, strideB=0, strideC=None, alpha=1.0, beta=0.0, workspace=None):
    lda = k if lda is None else lda
    strideA = m * lda if strideA is None else strideA
    strideC = m * n if strideC is None else strideC
    if workspace is not None and _sfx(A) in ("f16", "bf16"):
        fn = getattr(lib(), "sm_gemm_rowmajor_%s_ws" % _sfx(A))
        _check(fn(_dev(A), _dev(B), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC, float(alpha), float(beta), _dev(workspace),
                  workspace.numel() * workspace.element_size(), _stream()), "sm_gemm_rowmajor_ws")
        return
    fn = getattr(lib(), "sm_gemm_rowmajor_" + _sfx(A))
    _check(fn(_dev(A), _dev(B), _dev(C), m, n, k, lda, batch, strideA, strideB, strideC, float(alpha), float(beta),
              _stream()), "sm_gemm_rowmajor")


def copy_bytes(src, dst):
    """Streaming device-to-device copy of src's bytes into dst (bandwidth yardstick of bench.py)."""
    nbytes = src.numel() * src.element_size()
    if dst.numel() * dst.element_size() < nbytes:
        raise SparsifymeError("copy_bytes: destination smaller than the source")
    _check(lib().sm_copy_bytes(_dev(src), _dev(dst), nbytes, _stream()), "sm_copy_bytes")


def fill_uniform(out, seed, lo=0.0, hi=1.0):
    fn = getattr(lib(), "sm_fill_uniform_" + _sfx(out))
    _check(fn(_dev(out), out.numel(), seed, float(lo), float(hi), _stream()), "sm_fill_uniform")
