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


CONTRACT_LINE_MAX_BYTES = 4096   # BENCH_r05.json: a 20 KB line came back `parsed: null`; 14-18 KB lines (r02-r04) parsed.  Stay far below.


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def contract_line(out):
    """The ONE stdout line of bench.py: the contract's fields + `roofline` + `cpu_baseline` + a handful of stage scalars, as a
    compact JSON object of < CONTRACT_LINE_MAX_BYTES bytes.  Everything else the run measured (`stages` in full, `families`,
    `yardstick`, `verified_layers`, the per-shape tables) stays in the detail object, which bench.py writes to a FILE
    (--detail, default gpurun_out/bench_detail.json).  Pure function of the detail object: tests/test_bench_line.py feeds it a
    synthetic round-5-sized `out` and checks the size and the fields."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "partition_mode",
            "vs_baseline", "dtype", "data", "verified", "predicted")
    line = {k: _r(out[k]) for k in keep if k in out}
    cfg = out.get("config", {})
    lib = cfg.get("library") or {}
    line["config"] = {"workload": cfg.get("workload"), "path": (cfg.get("path") or "")[:200], "layers": cfg.get("layers"), "batch": cfg.get("batch"),
                      "dense_equiv_gflop_per_step": _r(cfg.get("dense_equiv_gflop_per_step"), 1), "launch": (cfg.get("launch") or "")[:60],
                      "streams": cfg.get("streams"), "partition": cfg.get("partition"), "layers_this_rank": cfg.get("layers_this_rank"), "parallelism": (cfg.get("parallelism") or "")[:120],
                      "library": {"sm_version": lib.get("sm_version"), "sha256_16": lib.get("sha256_16")},
                      "plan_costs": ((cfg.get("plan_costs") or {}).get("source") or "")[:80]}
    rf = out.get("roofline")
    if rf:
        ts = rf.get("traffic_source") or {}
        line["roofline"] = {k: _r(rf.get(k), 5) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_per_step", "avg_launch_us",
                                                          "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "step_frac") if k in rf}
        line["roofline"]["traffic_source"] = {"file": ts.get("file"), "stale": ts.get("stale")} if ts else None
        fams = rf.get("families") or {}
        if fams:   # one number per family: its fraction of the HBM peak (the full rows are in the detail file)
            line["roofline"]["families_frac"] = {n_.replace("spmma_f16_fused_", "").replace("spmma_f32_", "f32_"): _r(f_.get("frac_of_hbm_peak"), 3) for n_, f_ in fams.items()}
        ys = rf.get("yardstick") or {}
        if ys:
            line["roofline"]["device_copy_GBs"] = _r(ys.get("device_copy_GBs"), 1)
            line["roofline"]["step_GBs"] = _r(ys.get("step_GBs"), 1)
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value"), 2), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "dense_value": _r(cb.get("dense_value"), 2), "sample": (cb.get("sample") or "")[:260]}
    st = out.get("stages") or {}
    scal = ("spmma_mul_ms", "spmma_mul_grouped_ms", "compress_ms", "dense_gemm_rowmajor_ms", "dense_gemm_rowmajor_grouped_ms", "dense_gemm_batched_colmajor_ms",
            "speedup_full_vs_dense_rowmajor_grouped", "speedup_mul_grouped_vs_dense_rowmajor_grouped", "speedup_mul_vs_dense_batched",
            "hbm_bound_speedup_ceiling", "api_spmma_ms", "api_spmma_one_kernel_ms", "api_spmma_one_kernel_layers", "api_spmma_no_blob_layers", "api_spmma_frac_of_byte_floor")
    sline = {k: _r(st[k]) for k in scal if st.get(k) is not None}
    if isinstance(st.get("conv_step"), dict):
        sline["conv_step_ms"] = _r(st["conv_step"].get("conv_step_ms"))
    if isinstance(st.get("f32_split"), dict):
        sline["f32_split"] = {k: _r(v) for k, v in st["f32_split"].items() if k.endswith("_ms") and isinstance(v, float)}
    if isinstance(st.get("config4_sweep"), dict):
        c4 = st["config4_sweep"]
        sline["config4_sweep"] = {k: _r(c4.get(k)) for k in ("ms_per_step", "value", "unit", "n_gpus", "partition_mode")}
    if isinstance(st.get("matmul_mfma"), dict):
        sline["matmul_mfma_frac"] = _r(st["matmul_mfma"].get("frac"))
    line["stages"] = sline
    if "emulated" in out:
        line["emulated"] = out["emulated"]
    if out.get("detail_file"):
        line["detail_file"] = out["detail_file"]
    import json
    s = json.dumps(line, separators=(",", ":"))
    if len(s) > CONTRACT_LINE_MAX_BYTES:   # never again an unparseable headline: drop the optional blocks, in this order
        for k in ("stages", "emulated"):
            line.pop(k, None)
            s = json.dumps(line, separators=(",", ":"))
            if len(s) <= CONTRACT_LINE_MAX_BYTES:
                break
    return s
