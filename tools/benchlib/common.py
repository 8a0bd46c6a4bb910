"""Constants and helpers shared by bench.py and its stage modules."""
import csv
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH_PY = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
GUIDE_COPY_GBS = 6290.0    # the same guide, line 36: a float4 device copy measured at 6.29 TB/s (79 % of the specification)
F32_MATRIX_PEAK_TFS = 157.3  # same guide: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD


def read_shapes(path):
    with open(path, newline="") as f:
        rows = list(csv.reader(f))[1:]
    return [tuple(int(x) for x in r[:4]) for r in rows if r]


def table_path(name):
    if os.path.exists(name):
        return name
    p = os.path.join(ROOT, "datasets", name if name.endswith(".csv") else name + ".csv")
    if not os.path.exists(p):
        raise SystemExit(f"bench: no shape table {name!r}")
    return p


def file_tag(path, measured_on=None, loaded=None):
    """provenance of a replayed (not measured-in-this-run) profile file: relative path + content hash, the library it was
    measured on (tools/pmc_*.py record it) and whether that is the library THIS process loaded: `stale` = it is not (or the
    file does not say), and the caller then drops the replayed numbers instead of reporting another build's counters"""
    with open(path, "rb") as fh:
        return {"file": os.path.relpath(path, ROOT), "sha256_12": hashlib.sha256(fh.read()).hexdigest()[:12],
                "measured_in_this_run": False, "measured_on_library_sha256_16": measured_on, "loaded_library_sha256_16": loaded,
                "stale": (measured_on is None) or (measured_on != loaded),
                "how": "rocprofv3 --pmc passes of an earlier run of the same step (tools/pmc_traffic.py, tools/pmc_mfma.py); "
                       "the committed file is replayed here, the counters are not collected by bench.py itself"}


def library_tag(sm):
    """Which shared library this process loaded (SPARSIFYME_LIB can redirect it): path, version string, content hash."""
    with open(sm.LIB_PATH, "rb") as fh:
        h = hashlib.sha256(fh.read()).hexdigest()[:16]
    return {"lib_path": os.path.relpath(sm.LIB_PATH, ROOT) if sm.LIB_PATH.startswith(ROOT) else sm.LIB_PATH,
            "sm_version": sm.version(), "sha256_16": h, "redirected_by_env": bool(os.environ.get("SPARSIFYME_LIB"))}


def fused_variant(n, k, m=None, b=None, count=1, cus=256):
    """Which kernel sm_spmma_fused_f16[_grouped] dispatches a layer to (csrc/spmma_f16_fused.hip: spmma_fused16).  With m, b and
    the instance count of the launch given, the round-4 rule for the 256-row big form is applied too (it depends on how many
    tiles the launch has); without them the (n, k)-only families of rounds 1-3 are returned."""
    if n < 8 and k <= 64:
        return "thin"
    if k % 64 != 0:
        return "span"
    if n <= 128 or (n <= 256 and k <= 64):
        return "direct"
    astat = n > 256 and k <= 512
    if m is not None:
        rows = m * b                       # the batches of a shared-B launch are one tall matrix
        eff = lambda t: t / (-(-t // cus) * cus)
        t_big = -(-rows // 256) * -(-n // 256) * count
        t_wide = -(-rows // 128) * -(-n // 256) * count
        big = eff(t_big) >= eff(t_wide)
        if astat:
            panels, ns, tn = -(-rows // 128) * count, 1, -(-n // 128)
            while panels * ns * 4 < 3 * cus and -(-tn // (2 * ns)) >= 2:
                ns *= 2
            big = eff(t_big) > eff(panels * ns) + 0.1
        if big:
            return "big"
    return "astat" if astat else "wide"


def ge_mod():
    import __graft_entry__ as ge
    return ge
