"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU and
exports every symbol include/sparsifyme.h declares; host-only entry points behave; no compute call
is made here (that is what -m gpu is for)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sparsifyme.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds(pkg):
    assert declared_symbols() == pkg.EXPORTED_SYMBOLS


def test_library_builds_loads_and_exports_every_symbol(pkg):
    pkg.build()
    L = ctypes.CDLL(pkg.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), f"libsparsifyme.so does not export {name}"
    assert b"gfx950" in pkg.lib().sm_version()


def test_library_contains_gfx950_code_only(pkg):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={pkg.LIB_PATH}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets)
    strs = subprocess.run(["strings", "-n", "6", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in strs and "gfx942" not in strs and "sm_80" not in strs


def test_compress_size_host_entry_matches_oracle(pkg, orc):
    for (m, k, elt, b) in [(1, 4, 2, 1), (5, 147, 2, 3), (12544, 576, 2, 32), (196, 4608, 4, 32), (784, 64, 4, 1)]:
        assert pkg.compress24_size(m, k, elt, b) == orc.compress24_size(m, k, elt, b)
    out = ctypes.c_size_t(0)
    assert pkg.lib().sm_compress24_size(4, 4, 3, 1, ctypes.byref(out)) != 0       # bad element size
    assert b"invalid" in pkg.lib().sm_last_error()


def test_conv_out_size_host_entry(pkg, orc):
    """sm_conv_out_size is host arithmetic (datasets/get_shapes.py:19-20): same answers as the oracle and as the formula."""
    import math
    for (size, k, s, p, d) in [(224, 7, 2, 3, 1), (112, 3, 2, 1, 1), (56, 1, 1, 0, 1), (10, 3, 2, 0, 2), (5, 5, 1, 0, 1), (7, 3, 1, 1, 1)]:
        want = math.floor((size + 2 * p - d * (k - 1) - 1) / s + 1)
        assert pkg.conv_out_size(size, k, s, p, d) == orc.conv_out_size(size, k, s, p, d) == want
    out = ctypes.c_size_t(0)
    assert pkg.lib().sm_conv_out_size(4, 5, 1, 0, 1, ctypes.byref(out)) != 0      # window larger than the padded input
    assert pkg.lib().sm_conv_out_size(4, 3, 0, 0, 1, ctypes.byref(out)) != 0      # zero stride


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the product package or include/ may name it."""
    bad = []
    for base in (os.path.join(ROOT, "sparsify.me_amd"), os.path.join(ROOT, "include"), os.path.join(ROOT, "examples")):
        for dp, _, fns in os.walk(base):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h", ".hxx", ".cpp", ".inc", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    # comments may cite the oracle as the specification; code may not reach it
                    if re.search(r"libsm_oracle|#include\s+[\"<][^\">]*oracle|import\s+sm_oracle|load_oracle|"
                                 r"sm_oracle\.py|_ref\s*\(", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_missing_library_fails_loudly(pkg, tmp_path, monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("sparsifyme_amd_copy", os.path.join(ROOT, "sparsify.me_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.LIB_PATH = str(tmp_path / "nope.so")
    try:
        mod.lib()
    except mod.SparsifymeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("a missing HIP library must raise")


def test_coo_fast_form_rule_and_workspace_are_host_logic(pkg):
    """sm_spmm_coo_fast_form / sm_spmm_coo_fast_workspace_size run on the host: which matrix-core form a strided_coo call with the fast option
    gets (include/sparsifyme.h: the sparse matrix instruction for k <= 128, for matrices of at most 256 rows and for every k the dense-MFMA
    pipeline cannot take; beta != 0 and dense A: the pipeline or nothing), and a workspace that covers both layouts."""
    L = pkg.lib()
    form = lambda m, k, nnz, n, b, beta: L.sm_spmm_coo_fast_form(m, k, nnz, n, b, ctypes.c_float(beta))
    d = lambda m, k: m * k // 10
    assert form(12544, 64, d(12544, 64), 64, 32, 0.0) == 2          # k <= 128
    assert form(3136, 128, d(3136, 128), 512, 32, 0.0) == 2
    assert form(12544, 147, d(12544, 147), 64, 32, 0.0) == 2        # ragged k: only the sparse-instruction form takes it
    assert form(12544, 147, d(12544, 147), 64, 32, 0.5) == 0        # ... and only with beta == 0
    assert form(196, 4608, d(196, 4608), 512, 32, 0.0) == 2         # at most 256 rows
    assert form(3136, 1152, d(3136, 1152), 128, 32, 0.0) == 1       # many rows, long K: the dense-MFMA pipeline
    assert form(3136, 1152, d(3136, 1152), 128, 32, 1.0) == 1
    assert form(12544, 64, 12544 * 64 // 3, 64, 32, 0.0) == 1       # denser than 20 %
    assert form(130, 64, 100, 64, 2, 0.0) == 0                      # rows % 4 != 0: neither
    assert form(0, 64, 0, 64, 2, 0.0) == 0 and form(128, 64, 10, 0, 2, 0.0) == 0
    for (m, k, n, b) in ((12544, 576, 64, 32), (196, 4608, 512, 32), (8, 64, 8, 1), (1000, 1148, 16, 2)):
        nb = ctypes.c_size_t(0)
        assert L.sm_spmm_coo_fast_workspace_size(m, k, n, b, ctypes.byref(nb)) == 0
        kc, nst, tm = -(-k // 64) * 64, -(-k // 64), -(-m // 128)
        ru = lambda x: -(-x // 256) * 256
        sparse_layout = 256 + ru(m * kc * 4) + 2 * ru(nst * m * 64) + ru(nst * m * 8) + tm * nst * 256 * 8
        dense_layout = 256 + ru(n * b * k * 2) + 2 * ru(k * m * 4)
        assert nb.value >= max(sparse_layout, dense_layout)
