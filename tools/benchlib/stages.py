"""Rank 0's passes beside the timed step: per-stage times, the dense denominators, the API-faithful sequence, the roofline of the dominant kernel
family (`extras`), the convolution route, config 5 (COO) and the Blocked-ELL operator."""
import ctypes
import json
import os
import sys

from .cpu import config1_cpu
from .common import F32_MATRIX_PEAK_TFS, GUIDE_COPY_GBS, HBM_PEAK_GBS, ROOT, file_tag, fused_variant, ge_mod, read_shapes, table_path


def extras(args, sm, torch, dev, layers, flops, t_full, Forked, make_runner, timed, event_seconds, use_fused, out, grouping=None, step_full=None):
    """Rank 0 only: per-stage times, the dense denominators, the API-faithful sequence and the roofline of the
    dominant kernel family."""
    f32 = args.dtype == "f32"
    s = 4 if f32 else 2
    R = max(5, args.steps)

    def sec_per_call(fn):
        return timed(make_runner(fn), R, 2, collective=False) / R

    gfs = lambda t: flops / t / 1e9
    spmma_only = Forked(lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0))
    compress_only = Forked(lambda L: sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]))
    dense_rowmajor = Forked(lambda L: sm.gemm_rowmajor(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]))

    # the reference's dense path: column-major pointer-array batched GEMM, B shared (examples/gemm.cu:60,86)
    for L in layers:
        m, n, k, b = L["m"], L["n"], L["k"], L["b"]
        L["Ap"] = torch.tensor([L["A"].data_ptr() + s * i * m * k for i in range(b)], dtype=torch.int64, device=dev)
        L["Bp"] = torch.tensor([L["B"].data_ptr()] * b, dtype=torch.int64, device=dev)
        L["Cp"] = torch.tensor([L["C"].data_ptr() + s * i * m * n for i in range(b)], dtype=torch.int64, device=dev)
    has_batched = args.dtype in ("f16", "f32")  # cublas{H,S}gemmBatched's role; no bf16 form in the reference
    dense_batched = Forked(lambda L: sm.gemm_batched(L["Ap"], L["Bp"], L["Cp"], L["m"], L["n"], L["k"], L["b"], args.dtype))

    t_mul, t_cmp = sec_per_call(spmma_only), sec_per_call(compress_only)
    t_drm = sec_per_call(dense_rowmajor)
    # The dense comparator given the treatment the timed step gets (--group on): the instances of one shape as ONE grid.  The
    # row-major product of the (b*m) x k stacked operand IS the column-major pointer-array entry with the operands swapped
    # (C^T = B^T A^T: the same kernel, sm_gemm_batched_* maps it back), so a group is one call with `count` pointer triples.
    t_drm_grouped = None
    if grouping and has_batched:
        fused_groups, _, spread, ForkedItems = grouping[:4]
        ditems = []
        dense_ws = {}   # the dense twin gets the stream-K workspace too (sm_gemm_batched_f16_ws): the library's rule decides, as for the 2:4 launches
        for _, Ls in fused_groups(layers):
            if len(Ls) == 1:
                ditems.append(("single", Ls))
                continue
            for L in Ls[:1]:
                L["gAp"] = torch.tensor([x["A"].data_ptr() for x in Ls], dtype=torch.int64, device=dev)
                L["gBp"] = torch.tensor([x["B"].data_ptr() for x in Ls], dtype=torch.int64, device=dev)
                L["gCp"] = torch.tensor([x["C"].data_ptr() for x in Ls], dtype=torch.int64, device=dev)
            ditems.append(("group", Ls))

        def dense_group(Ls):
            L0 = Ls[0]
            key = (L0["m"], L0["n"], L0["k"])
            if key not in dense_ws:
                dense_ws[key] = sm.spmma_fused_workspace() if (args.streamk == "on" and args.dtype == "f16" and 128 < L0["n"] <= 256 and L0["k"] >= 2048) else None
            sm.gemm_batched(L0["gBp"], L0["gAp"], L0["gCp"], L0["n"], L0["m"] * L0["b"], L0["k"], len(Ls), args.dtype, workspace=dense_ws[key])
        t_drm_grouped = sec_per_call(ForkedItems(spread(ditems), dense_group,
                                                 lambda L: sm.gemm_rowmajor(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"])))
    # the 2:4 matmul on prepared blobs given the same treatment (round 4: sm_spmma_*_grouped, one grid per <= 8 same-shape blobs)
    t_mul_grouped = None
    if grouping and not f32 and hasattr(sm, "spmma_grouped"):
        fused_groups, _, spread, ForkedItems = grouping[:4]
        mitems = [("group" if len(Ls) > 1 else "single", Ls) for _, Ls in fused_groups(layers)]

        def mul_group(Ls):
            L0 = Ls[0]
            sm.spmma_grouped([x["blob"] for x in Ls], [x["B"] for x in Ls], [x["C"] for x in Ls], L0["m"], L0["n"], L0["k"], batch=L0["b"])
        t_mul_grouped = sec_per_call(ForkedItems(spread(mitems), mul_group,
                                                 lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)))
    t_dcm = sec_per_call(dense_batched) if has_batched else None
    t_staged = t_full if args.path == "staged" else sec_per_call(Forked(lambda L: (
        sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
        sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0))))
    prev_stages = out.get("stages", {})
    out["stages"] = {
        "spmma_mul_gfs": gfs(t_mul), "spmma_mul_ms": t_mul * 1e3, "compress_ms": t_cmp * 1e3,
        "dense_gemm_rowmajor_gfs": gfs(t_drm), "dense_gemm_rowmajor_ms": t_drm * 1e3,
        "dense_gemm_batched_colmajor_gfs": gfs(t_dcm) if t_dcm else None, "dense_gemm_batched_colmajor_ms": t_dcm * 1e3 if t_dcm else None,
        "speedup_mul_vs_dense_rowmajor": t_drm / t_mul, "speedup_mul_vs_dense_batched": t_dcm / t_mul if t_dcm else None,
        "speedup_full_vs_dense_rowmajor": t_drm / t_full, "speedup_full_vs_dense_batched": t_dcm / t_full if t_dcm else None,
        "dense_gemm_rowmajor_grouped_ms": t_drm_grouped * 1e3 if t_drm_grouped else None,
        "dense_gemm_rowmajor_grouped_gfs": gfs(t_drm_grouped) if t_drm_grouped else None,
        "speedup_full_vs_dense_rowmajor_grouped": t_drm_grouped / t_full if t_drm_grouped else None,
        "speedup_mul_vs_dense_rowmajor_grouped": t_drm_grouped / t_mul if t_drm_grouped else None,
        "spmma_mul_grouped_ms": t_mul_grouped * 1e3 if t_mul_grouped else None,
        "spmma_mul_grouped_gfs": gfs(t_mul_grouped) if t_mul_grouped else None,
        "speedup_mul_grouped_vs_dense_rowmajor_grouped": t_drm_grouped / t_mul_grouped if t_drm_grouped and t_mul_grouped else None,
        "speedup_mul_grouped_vs_dense_batched": t_dcm / t_mul_grouped if t_dcm and t_mul_grouped else None,
        "full_path_staged_gfs": gfs(t_staged), "full_path_staged_ms": t_staged * 1e3,
        "timed_path": args.path, "timed_path_ms": t_full * 1e3,
        # what 2:4 can buy on these shapes when both products are HBM-bound (fp16: they are): the ratio of
        # the algorithmic bytes, dense (A + B + C) over sparse (9/16 A + B + C)
        "hbm_bound_speedup_ceiling": sum(L["b"] * s * (L["m"] * L["k"] + L["m"] * L["n"]) + s * L["k"] * L["n"] for L in layers)
        / sum(L["b"] * (L["m"] * L["k"] * (s / 2 + 1.0 / 8) + s * L["m"] * L["n"]) + s * L["k"] * L["n"] for L in layers),
    }
    out["stages"].update(prev_stages)

    # What the per-step join costs (NOT the headline: `value` keeps one fork / join per step).  The steps are independent
    # batches; here four of them are captured as one graph in which every stream runs its chain four times back to back --
    # a layer's consecutive executions stay ordered on their stream, nothing waits for another stream between steps -- so one
    # step's ramp-up and tail (<= 2 kernels active for ~ 25 % of a replayed step, profiles/ktrace_r03f.txt) overlap its neighbours.
    if step_full is not None and not args.eager:
        class Pipelined(object):
            def __call__(self):
                step_full.fork_join(lambda w: [step_full.chain(w) for _ in range(4)])
        t_pipe = timed(make_runner(Pipelined()), max(2, R // 2), 2, collective=False) / max(2, R // 2) / 4.0
        out["stages"]["pipelined_4_steps_ms_per_step"] = t_pipe * 1e3
        out["stages"]["pipelined_4_steps_gfs"] = gfs(t_pipe)
        out["stages"]["pipelined_note"] = ("four steps per graph replay, no cross-stream join between them (per-stream order kept); "
                                           "reported beside the headline, which joins every step")

    # The API-faithful sequence of sparsifyme::spmma() (reference spmma.hxx:82-113, include/sparsify.me/spmma.hxx):
    # TILE prune -> prune check -> compress -> multiply.  The prune reads the step's dense A and writes the pruned
    # operand to a second buffer (the bytes of the in-place prune, without turning the bench's operand into an already
    # pruned one for the next step).
    if hasattr(sm, "api_spmma_step"):  # (fp32 too since round 3: sm_prune24_compress24_f32)
        valid = torch.zeros(1, dtype=torch.int32, device=dev)
        for L in layers:
            L["Aapi"] = torch.empty_like(L["A"])
        t_api = sec_per_call(Forked(lambda L: sm.api_spmma_step(L["A"], L["Aapi"], L["B"], L["C"], L["blob"], valid, L["m"], L["n"], L["k"], L["b"])))
        out["stages"]["api_spmma_ms"] = t_api * 1e3
        out["stages"]["api_spmma_gfs"] = gfs(t_api)
        out["stages"]["api_spmma_sequence"] = sm.API_SPMMA_SEQUENCE
        if not f32 and hasattr(sm, "api_spmma_step_fused"):
            # rounds 4 + 6: the same sequence WITHOUT a blob through sm_prune24_spmma_*: ONE kernel (TILE prune written to the second buffer, flag,
            # multiply) on the layers it takes (n <= 128, k % 64 == 0, m % 4 == 0); on every other layer the exact fused kernels take (round 6: all
            # of ResNet-50) the prune + flag pass over A followed by the fused kernel on the pruned operand; the blob pair only where neither applies
            taken = {}

            def api_no_blob(L):
                rc = sm.prune24_spmma(L["A"], L["Aapi"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"], d_valid=valid, check=False)
                taken[L["li"]] = rc == 0
                if rc != 0:
                    sm.api_spmma_step(L["A"], L["Aapi"], L["B"], L["C"], L["blob"], valid, L["m"], L["n"], L["k"], L["b"])
            t_api1 = sec_per_call(Forked(api_no_blob))
            n_one = sum(1 for L in layers if L["n"] <= 128 and L["n"] % 8 == 0 and L["k"] % 64 == 0 and L["m"] % 4 == 0)
            out["stages"]["api_spmma_one_kernel_ms"] = t_api1 * 1e3
            out["stages"]["api_spmma_one_kernel_gfs"] = gfs(t_api1)
            out["stages"]["api_spmma_one_kernel_layers"] = n_one
            out["stages"]["api_spmma_no_blob_layers"] = sum(1 for v in taken.values() if v)
            out["stages"]["api_spmma_no_blob_sequence"] = ("sm_prune24_spmma_*: one kernel on %d layers; prune + flag pass (sm_prune24_compress24_* with a null blob) + "
                                                           "sm_spmma_fused_* on the pruned operand on %d; blob pair on %d" % (n_one, sum(1 for v in taken.values() if v) - n_one,
                                                                                                                            sum(1 for v in taken.values() if not v)))
            # the byte floor of the sequence with A pruned in place: 2 A + C + B at the HBM peak
            floor_b = sum(L["b"] * s * (2 * L["m"] * L["k"] + L["m"] * L["n"]) + s * L["k"] * L["n"] for L in layers)
            out["stages"]["api_spmma_byte_floor_ms"] = floor_b / (HBM_PEAK_GBS * 1e9) * 1e3
            out["stages"]["api_spmma_frac_of_byte_floor"] = floor_b / (HBM_PEAK_GBS * 1e9) / t_api1
        for L in layers:
            del L["Aapi"]

    if f32 and hasattr(sm, "spmma_fused_f32_split"):
        # round 4: the fp32 2:4 product on the SPARSE matrix instruction through exact bfloat16 splits of both operands
        # (sm_spmma_fused_f32_split; planes = 3: |error| <= 2^-21 sum|a||b|, planes = 2: 2^-13) where it applies (k % 64 == 0,
        # n % 8 == 0), the exact fused kernel elsewhere.  Reported BESIDE the headline, which stays the exact fp32 form.
        split = {"kernel": "spmma_f32_split_kernel (v_smfmac_f32_16x16x64_bf16 on three / two truncated bfloat16 pieces per fp32 value, fp32 "
                           "accumulation; mask = the exact path's) + split_planes_kernel (B's pieces, once per call, into a workspace)"}
        for L in layers:
            L["ws"] = torch.empty(max(16, sm.spmma_fused_f32_split_workspace(L["n"], L["k"], planes=3)), dtype=torch.uint8, device=dev)
        # the exact fp32 form (the headline until round 5; --f32-planes 0), timed here when the step itself runs the split form
        def layer_exact(L):
            (sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]) if use_fused(L) else
             (sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
              sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)))
        t_exact = sec_per_call(Forked(layer_exact)) if any(L.get("split") for L in layers) else t_full
        split["exact_fused_ms"] = t_exact * 1e3
        split["exact_fused_speedup_vs_dense_rowmajor"] = t_drm / t_exact
        split["timed_path_speedup_vs_dense_rowmajor"] = t_drm / t_full
        for planes in (3, 2):
            def layer_split(L, planes=planes):
                if sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes, check=False) != 0:
                    (sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]) if use_fused(L) else
                     (sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                      sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)))
            t_sp = sec_per_call(Forked(layer_split))
            key = "planes%d" % planes
            split[key + "_ms"] = t_sp * 1e3
            split[key + "_gfs"] = gfs(t_sp)
            split[key + "_speedup_vs_dense_rowmajor"] = t_drm / t_sp
            split[key + "_speedup_vs_exact_fused"] = t_exact / t_sp
            split[key + "_hbm_frac"] = sum(L["b"] * 4 * (L["m"] * L["k"] + L["m"] * L["n"]) + 4 * L["k"] * L["n"] for L in layers) / t_sp / (HBM_PEAK_GBS * 1e9)
        # B's planes kept across calls (round 5: sm_spmma_fused_f32_split_prepare once, untimed -- B is the layer's weights in the reference's
        # use -- then sm_spmma_fused_f32_split_prepared per step): the same C bit for bit, without the per-call pass over B
        if hasattr(sm, "spmma_fused_f32_split_prepared"):
            for planes in (3, 2):
                for L in layers:
                    L["prep"] = sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes, check=False) == 0
                    if L["prep"]:
                        sm.spmma_fused_f32_split_prepare(L["B"], L["n"], L["k"], L["ws"], planes=planes)
                def layer_prepared(L, planes=planes):
                    if L["prep"]:
                        sm.spmma_fused_f32_split_prepared(L["A"], L["ws"], L["C"], L["m"], L["n"], L["k"], batch=L["b"], planes=planes)
                    else:
                        (sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]) if use_fused(L) else
                         (sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                          sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)))
                t_pp = sec_per_call(Forked(layer_prepared))
                split["planes%d_prepared_ms" % planes] = t_pp * 1e3
                split["planes%d_prepared_speedup_vs_dense_rowmajor" % planes] = t_drm / t_pp
        # the dense product by the same pieces (sm_gemm_rowmajor_f32_split): what the 2:4 split form should be held against
        for planes in (3, 2):
            def layer_dense_split(L, planes=planes):
                if sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes, check=False, dense=True) != 0:
                    sm.gemm_rowmajor(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"])
            t_ds = sec_per_call(Forked(layer_dense_split))
            split["dense_planes%d_ms" % planes] = t_ds * 1e3
            split["planes%d_speedup_vs_dense_split" % planes] = t_ds / (split["planes%d_ms" % planes] * 1e-3)
        # the API-faithful sequence of spmma<float> with spmma_options().f32_planes: TILE prune in place (here: into a second buffer, as
        # stages.api_spmma_ms does) + check in one pass, no blob, then the split multiply straight from the pruned dense operand
        for L in layers:
            L["Aapi"] = torch.empty_like(L["A"])
        vflag = torch.zeros(1, dtype=torch.int32, device=dev)
        for planes in (3, 2):
            def layer_api_split(L, planes=planes):
                sm.prune24_compress24(L["A"], L["Aapi"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], None, vflag, sm.PRUNE_TILE)
                if sm.spmma_fused_f32_split(L["Aapi"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes, check=False) != 0:
                    sm.compress24(L["Aapi"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"])
                    sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)
            split["api_spmma_planes%d_ms" % planes] = sec_per_call(Forked(layer_api_split)) * 1e3
        for L in layers:
            del L["Aapi"]
        split["layers_on_split_form"] = sum(1 for L in layers if sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"],
                                                                                      planes=3, check=False) == 0)
        # error of the split forms against the exact kernel on the first layer they take (max |diff| / max sum|a||b| bound proxy)
        L = next((L for L in layers if L["k"] % 64 == 0 and L["n"] % 8 == 0), None)
        if L is not None:
            Ce = torch.empty_like(L["C"])
            sm.spmma_fused(L["A"], L["B"], Ce, L["m"], L["n"], L["k"], batch=L["b"])
            for planes in (3, 2):
                sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=planes)
                torch.cuda.synchronize()
                d = (L["C"].double() - Ce.double()).abs().max().item()
                split["planes%d_max_abs_diff_vs_exact" % planes] = d
                split["planes%d_max_rel_diff_vs_exact" % planes] = d / max(Ce.double().abs().max().item(), 1e-30)
            split["diff_layer"] = [L["m"], L["n"], L["k"], L["b"]]
            del Ce
        out["stages"]["f32_split"] = split

    if not f32 and os.path.basename(args.tables.split(",")[0] if args.tables else (args.table or "resnet50")).startswith("resnet50"):
        out["stages"]["conv_path"] = conv_path_stage(sm, torch, dev, args.dtype)
        if grouping and len(layers) == 49:
            out["stages"]["conv_step"] = conv_step_stage(args, sm, torch, dev, layers, grouping, make_runner, event_seconds, out["stages"])
        out["stages"]["config5_coo_spmm"] = config5_stage(sm, torch, dev)
        out["stages"]["bell_spmm"] = bell_stage(sm, torch, dev)
    if not args.no_cpu_baseline:
        out["stages"]["config1_cpu"] = config1_cpu(ge_mod())

    if not f32:
        # matrix-pipe view of the 2:4 matmul (north_star: "MFMA utilisation for the matmul against chip peak"):
        # dense-equivalent rate of the matmul-only pass against 2 x the dense fp16 peak (v_smfmac does a 16x16x64
        # product in the cycles of a dense 16x16x32), plus the PMC MfmaUtil per kernel when a profile is present
        mfma = {"achieved_TFs": gfs(t_mul) / 1e3, "peak_TFs": 2.0 * 2500.0, "frac": gfs(t_mul) / 1e3 / 5000.0,
                "peak": "2 x 2.5 PF/s dense fp16 (MI355X_MICROARCH.md); the v_smfmac issue rate measured on this chip "
                        "is 3.4-3.8 PF/s dense-equivalent (profiles/mfma_rate_r01.txt)",
                "pmc_mfma_util_percent": None, "pmc_source": None}
        mpath = os.path.join(ROOT, "profiles", "mfma_util_latest.json")  # tools/pmc_mfma.py, from a rocprofv3 --pmc pass
        if os.path.exists(mpath):
            try:
                mtab = json.load(open(mpath))
                mfma["pmc_source"] = file_tag(mpath, (mtab.get("_library") or {}).get("sha256_16"), out["config"]["library"]["sha256_16"])
                if not mfma["pmc_source"]["stale"]:   # counters of another build are not this build's utilisation
                    mfma["pmc_mfma_util_percent"] = {k: round(v["mfma_util_percent"], 2) for k, v in mtab.items() if not k.startswith("_")}
            except Exception:
                pass
        out["stages"]["matmul_mfma"] = mfma

    # roofline of the dominant kernel family of the timed step: algorithmic bytes / device
    # time (HIP events on the launch stream) of a single-stream pass that launches only that family on its layers
    A_sp = lambda L: L["b"] * (L["m"] * L["k"] * s / 2 + L["m"] * L["k"] / 8 + L["m"] * L["n"] * s) + s * L["k"] * L["n"]
    A_fu = lambda L: L["b"] * s * (L["m"] * L["k"] + L["m"] * L["n"]) + s * L["k"] * L["n"]
    fam = {}
    if f32:
        staged = [L for L in layers if not use_fused(L)]
        fam["spmma_f32"] = dict(names=["spmma_f32_dma_kernel", "spmma_f32_kernel"], layers=staged,
                                call=lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0), bytes=A_sp)
        fam["compress"] = dict(names=["compress_kernel"], layers=staged,
                               call=lambda L: sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                               bytes=lambda L: L["b"] * L["m"] * L["k"] * (s + s / 2 + 1 / 8))
        fam["spmma_f32_fused"] = dict(names=["gemm_f32_dma_kernel"], layers=[L for L in layers if use_fused(L) and not L.get("split")],
                                      call=lambda L: sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]), bytes=A_fu)
        # round 6: the split form is the timed path by default (--f32-planes 3): one family per kernel it dispatches to.  Bytes = the dense fp32
        # A once + C + B (fp32, read by the per-call split pass) + B's planes written and read (planes x 2 bytes per element, each way)
        pl_ = getattr(args, "f32_planes", 0)
        A_split = lambda L: A_fu(L) + 2.0 * pl_ * 2 * L["k"] * L["n"]

        def split_kernel_of(L):
            if L["k"] % 64 != 0:
                return "span"
            return "cols" if 128 < L["n"] <= 256 else "tile"
        for var, kname in (("tile", "spmma_f32_split_kernel"), ("cols", "spmma_f32_split_cols_kernel"), ("span", "spmma_f32_split_span_kernel")):
            fam["spmma_f32_split_" + var] = dict(names=[kname], layers=[L for L in layers if L.get("split") and split_kernel_of(L) == var],
                                                 call=lambda L: sm.spmma_fused_f32_split(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], L["ws"], batch=L["b"], planes=pl_),
                                                 bytes=A_split)
        staged = [L for L in staged if not L.get("split")]
        fam["spmma_f32"]["layers"] = staged
        fam["compress"]["layers"] = staged
    else:
        staged = [L for L in layers if not use_fused(L)]
        fam["spmma_f16"] = dict(names=["spmma_f16_dma_kernel", "spmma_f16_pc_kernel", "spmma_f16_kernel", "spmma_f16_splitk_kernel"],
                                layers=staged, call=lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0), bytes=A_sp)
        fam["compress"] = dict(names=["compress_flat_kernel", "compress_rowspan_f16_kernel", "compress_kernel"], layers=staged,
                               call=lambda L: sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                               bytes=lambda L: L["b"] * L["m"] * L["k"] * (s + s / 2 + 1 / 8))
        shape_count = {}
        for L in layers:
            if use_fused(L):
                key = (L["m"], L["n"], L["k"], L["b"])
                shape_count[key] = shape_count.get(key, 0) + 1

        def variant_of(L):  # the kernel the timed step's (grouped) launch of this layer's shape runs
            cnt = min(8, shape_count[(L["m"], L["n"], L["k"], L["b"])]) if grouping else 1
            if grouping and grouping[4](L, cnt):   # the library's own rule (sm_spmma_fused_streamk_plan): the stream-K form
                return "sk"
            return fused_variant(L["n"], L["k"], L["m"], L["b"], cnt)
        for var in ("direct", "big", "wide", "astat", "span", "sk", "thin"):
            fam["spmma_f16_fused_" + var] = dict(names=["spmma_f16_thin_kernel"] if var == "thin" else ["spmma_f16_fused_%s_kernel" % var] + (["spmma_f16_fused_widep_kernel"] if var == "wide" else []),
                                                 layers=[L for L in layers if use_fused(L) and variant_of(L) == var],
                                                 call=lambda L: sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]),
                                                 bytes=A_fu)
    traffic_tab, tsrc = {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic_f32_latest.json" if f32 else "traffic_latest.json")  # tools/pmc_traffic.py, from rocprofv3 --pmc passes
    if os.path.exists(tpath):
        try:
            traffic_tab = json.load(open(tpath))
            tsrc = file_tag(tpath, (traffic_tab.get("_library") or {}).get("sha256_16"), out["config"]["library"]["sha256_16"])
            if tsrc["stale"]:   # measured on another build of the library: not replayed (roofline.traffic stays null, the tag says why)
                traffic_tab = {}
        except Exception:
            traffic_tab = {}
    rows = {}
    for name, f in fam.items():
        if not f["layers"]:
            continue

        nlaunch = len(f["layers"])
        if grouping and name.startswith("spmma_f16_fused"):  # the family as the timed step launches it: one grid per <= 8 instances of a shape
            gl = grouping[0](f["layers"])
            nlaunch = sum((len(Ls) + 7) // 8 for _, Ls in gl)

            def serial(gl=gl):
                for _, Ls in gl:
                    grouping[1](Ls)
        else:
            def serial(f=f):
                for L in f["layers"]:
                    f["call"](L)
        t = event_seconds(make_runner(serial), R)
        by = sum(f["bytes"](L) for L in f["layers"])
        fl = sum(2.0 * L["m"] * L["n"] * L["k"] * L["b"] for L in f["layers"])
        tb = [(traffic_tab[n]["hbm_bytes_per_launch"], traffic_tab[n]["launches_profiled"]) for n in f["names"] if n in traffic_tab]
        traffic = sum(b_ * c_ for b_, c_ in tb) / sum(c_ for _, c_ in tb) if tb else None
        rows[name] = dict(seconds=t, launches=nlaunch, layers=len(f["layers"]), bytes=by, GBs=by / t / 1e9, TFs=fl / t / 1e12, traffic=traffic)
    # the fused variants are one family for the "dominant kernel" choice (they are one entry point), reported each
    groups = {}
    for n_, r_ in rows.items():
        groups.setdefault("spmma_f16_fused" if n_.startswith("spmma_f16_fused") else n_, []).append(r_)
    gsum = {g: dict(seconds=sum(r["seconds"] for r in rs), launches=sum(r["launches"] for r in rs), bytes=sum(r["bytes"] for r in rs),
                    TFs=None, traffic=(sum(r["traffic"] * r["launches"] for r in rs) / sum(r["launches"] for r in rs)
                                       if all(r["traffic"] is not None for r in rs) else None)) for g, rs in groups.items()}
    dom = max(gsum, key=lambda g: gsum[g]["seconds"])
    d = gsum[dom]
    fams_out = {n_: {"ms_per_step": r_["seconds"] * 1e3, "launches": r_["launches"], "layers": r_["layers"], "GBs": r_["GBs"], "frac_of_hbm_peak": r_["GBs"] / HBM_PEAK_GBS,
                     "hbm_traffic_per_launch": r_["traffic"]} for n_, r_ in rows.items()}
    if f32 and any(n_.startswith("spmma_f32_split") for n_ in rows):
        # the split form: bound by the HBM stream of the fp32 A (the 6 / 3 sparse bf16 instructions per block are 0.52 / 0.26 ms of matrix
        # time on ResNet-18 against 0.96 ms of bytes at the peak)
        domf = max((n_ for n_ in rows if n_.startswith("spmma_f32_split")), key=lambda n_: rows[n_]["seconds"])
        r_ = rows[domf]
        step_bytes = sum(rows[n_]["bytes"] for n_ in rows if n_ != "compress")
        out["roofline"] = {"bound": "hbm", "achieved": r_["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r_["GBs"] / HBM_PEAK_GBS, "traffic": r_["traffic"],
                           "traffic_source": tsrc if (r_["traffic"] is not None or (tsrc and tsrc["stale"])) else None,
                           "kernel": fam[domf]["names"][0], "launches_per_step": r_["launches"], "avg_launch_us": r_["seconds"] / r_["launches"] * 1e6,
                           "algorithmic_bytes_per_launch": r_["bytes"] / r_["launches"], "step_frac": step_bytes / t_full / 1e9 / HBM_PEAK_GBS,
                           "matrix_pipe": {"executed_TFs_dense_equivalent": r_["TFs"] * 2 * getattr(args, "f32_planes", 3), "peak_TFs": 5000.0,
                                           "note": "planes = 3: six v_smfmac_f32_16x16x64_bf16 per 16 x 16 x 64 block (three at planes = 2)"},
                           "measured": "single stream, one kernel family at a time, HIP events on the launch stream, hipGraph replay", "families": fams_out}
    elif f32:
        domf = max((n_ for n_ in rows if n_.startswith("spmma_f32")), key=lambda n_: rows[n_]["seconds"])
        r_ = rows[domf]
        out["roofline"] = {"bound": "mfma", "achieved": r_["TFs"], "peak": F32_MATRIX_PEAK_TFS, "unit": "TFLOP/s",
                           "frac": r_["TFs"] / F32_MATRIX_PEAK_TFS, "traffic": r_["traffic"], "traffic_source": tsrc if (r_["traffic"] is not None or (tsrc and tsrc["stale"])) else None,
                           "kernel": domf,
                           "launches_per_step": r_["launches"], "avg_launch_us": r_["seconds"] / r_["launches"] * 1e6,
                           "algorithmic_flops_per_launch": r_["TFs"] * 1e12 * r_["seconds"] / r_["launches"],
                           "note": "the fp32 2:4 kernel expands to dense fp32 MFMA (no fp32 sparse matrix instruction exists): executed = dense-equivalent flops",
                           "measured": "single stream, HIP events on the launch stream, hipGraph replay", "families": fams_out}
    else:
        # round 6: the block names ONE kernel -- the family the step spends most of its time in (direct) -- with that kernel's own
        # bytes, launches, event time and counter traffic; the five fused families together (one entry point) are `entry_point`
        domk = max(rows, key=lambda n_: rows[n_]["seconds"])
        rk = rows[domk]
        entry = {"name": dom, "GBs": d["bytes"] / d["seconds"] / 1e9, "frac": d["bytes"] / d["seconds"] / 1e9 / HBM_PEAK_GBS, "launches_per_step": d["launches"],
                 "avg_launch_us": d["seconds"] / d["launches"] * 1e6, "algorithmic_bytes_per_launch": d["bytes"] / d["launches"], "traffic": d["traffic"]}
        d = dict(seconds=rk["seconds"], launches=rk["launches"], bytes=rk["bytes"], traffic=rk["traffic"])
        dom = domk + "_kernel"
        GBs = d["bytes"] / d["seconds"] / 1e9
        # yardstick measured in THIS process: sm_copy_bytes (16-byte non-temporal loads + stores) moving the timed step's own
        # algorithmic byte count (half read, half written), same event timing as the families above
        step_bytes = sum((A_fu(L) if use_fused(L) else A_sp(L) + L["b"] * L["m"] * L["k"] * (s + s / 2 + 1 / 8)) for L in layers)
        half = int(step_bytes / 2) // 4096 * 4096
        ysrc = torch.empty(half, dtype=torch.uint8, device=dev)
        ydst = torch.empty(half, dtype=torch.uint8, device=dev)
        sm.fill_uniform(ysrc.view(torch.float16), 0xC0B1, 0.0, 1.0)
        t_copy = event_seconds(make_runner(lambda: sm.copy_bytes(ysrc, ydst)), R)
        copy_GBs = 2.0 * half / t_copy / 1e9
        del ysrc, ydst
        out["roofline"] = {"bound": "hbm", "achieved": GBs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": GBs / HBM_PEAK_GBS, "traffic": d["traffic"], "traffic_source": tsrc if (d["traffic"] is not None or (tsrc and tsrc["stale"])) else None,
                           "kernel": dom, "launches_per_step": d["launches"], "avg_launch_us": d["seconds"] / d["launches"] * 1e6,
                           "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                           "measured": "single stream, one kernel family at a time, HIP events on the launch stream, hipGraph replay",
                           "yardstick": {"device_copy_GBs": copy_GBs, "frac_of_device_copy": GBs / copy_GBs,
                                         "guide_float4_copy_GBs": GUIDE_COPY_GBS, "frac_of_guide_copy": GBs / GUIDE_COPY_GBS,
                                         "step_frac_of_guide_copy": step_bytes / t_full / 1e9 / GUIDE_COPY_GBS,
                                         "copy_bytes": 2 * half, "copy_ms": t_copy * 1e3,
                                         "step_algorithmic_bytes": step_bytes, "step_GBs": step_bytes / t_full / 1e9,
                                         "step_frac_of_device_copy": step_bytes / t_full / 1e9 / copy_GBs,
                                         "source": "measured in this run: sm_copy_bytes (round 5: one 16-byte non-temporal load + store per thread, no loop -- the "
                                                   "fastest of the copy forms of tools/probes/copy_probe.hip, 6.5-6.6 TB/s on random data where the grid-stride form used "
                                                   "until round 4 gave 5.3-5.9) over the timed step's own algorithmic byte count of random halves, half read + half written; "
                                                   "guide_float4_copy_GBs = the float4 copy /opt/skills/guides/MI355X_MICROARCH.md:36 measures (6.29 TB/s); context "
                                                   "only -- `frac` is against the 8 TB/s specification"},
                           "step_frac": step_bytes / t_full / 1e9 / HBM_PEAK_GBS, "entry_point": entry,
                           "families": fams_out}


def conv_path_stage(sm, torch, dev, dtype):
    """The 3 x 3 convolution layers of the ResNet-50 table through the implicit-GEMM kernel (sm_conv_spmma_fused_*: NCHW
    activations in, C out, neither the 9 x larger A nor its blob in HBM), stride 1 / padding 1 so that m = H * W, b = 32.
    Own roofline: bytes = activations + B + C; bound = max(bytes / HBM peak, dense-equivalent flops / 2 x 2.5 PF).  Not part
    of `value`: the headline step is defined on the reference's (m, n, k) operands."""
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    N = 32
    rows, tot_ms, tot_fl, tot_by, tot_roof, tot_routed = [], 0.0, 0.0, 0.0, 0.0, 0.0
    for Cin, HW, n, cnt in [(64, 112, 64, 3), (128, 56, 128, 4), (256, 28, 256, 6), (512, 14, 512, 3)]:
        L, K = HW * HW, Cin * 9
        X = torch.empty(N * Cin * L, dtype=tdt, device=dev)
        sm.fill_uniform(X, 7 + Cin, -1.0, 1.0)
        B = torch.empty(K * n, dtype=tdt, device=dev)
        sm.fill_uniform(B, 9 + n, -1.0, 1.0)
        C = torch.empty(N * L * n, dtype=tdt, device=dev)
        ms = sm.graph_time_ms(lambda: sm.conv_spmma_fused(X, B, C, N, Cin, HW, HW, 3, 3, 1, 1, 1, n), iters=10)
        fl, by = 2.0 * N * L * n * K, 2.0 * (N * Cin * L + K * n + N * L * n)
        roof = max(by / (HBM_PEAK_GBS * 1e9), fl / 5.0e15)
        row = {"m": L, "n": n, "k": K, "count": cnt, "ms": ms, "GBs": by / ms / 1e6, "eff_TFs": fl / ms / 1e9, "frac": roof * 1e3 / ms}
        # round 4: sm_conv_spmma_* picks the faster route per layer (small-spatial long-K layers: im2col-to-blob + staged matmul)
        ms_r = ms
        if hasattr(sm, "conv_spmma"):
            need = sm.conv_spmma_workspace(N, Cin, HW, HW, 3, 3, 1, 1, 1)
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
            ms_r = sm.graph_time_ms(lambda: sm.conv_spmma(X, B, C, N, Cin, HW, HW, 3, 3, 1, 1, 1, n, workspace=ws), iters=10)
            row["ms_routed"] = ms_r
            row["route"] = "im2col_compress24 + spmma" if need else "implicit GEMM"
            del ws
        rows.append(row)
        tot_ms += ms * cnt; tot_fl += fl * cnt; tot_by += by * cnt; tot_roof += roof * cnt; tot_routed += ms_r * cnt
        del X, C
    return {"kernel": "conv_spmma_fused_kernel", "layers": rows, "table_weighted_ms": tot_ms, "table_weighted_routed_ms": tot_routed,
            "eff_TFs": tot_fl / tot_ms / 1e9,
            "roofline": {"bound": "hbm", "achieved": tot_by / tot_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": tot_by / tot_ms / 1e6 / HBM_PEAK_GBS, "frac_of_per_layer_roofline": tot_roof * 1e3 / tot_ms},
            "note": "bytes = activations + B + C (no A); DESIGN.md 4.4: bound by the selection / gather instruction stream, not by HBM"}


def conv_step_stage(args, sm, torch, dev, layers, grouping, make_runner, event_seconds, stages):
    """The whole ResNet-50 table FROM ACTIVATIONS as one replayed step: the 33 1 x 1 layers are the fused
    kernel on their NHWC activations (which ARE the (m x k) operand: the timed step's own launches), the 16 3 x 3 layers and the
    7 x 7 stem go through sm_conv_spmma_* from NCHW activations -- implicit GEMM, or im2col-to-blob + matmul where the routing
    rule / the geometry says so -- so the kh x kw times larger A of those 17 layers is never materialised.  Same fork / join over
    the streams as the headline step, one hipGraph replay.  Verified AFTER the loop: every convolution layer's C against
    sm_im2col_compress24 + sm_spmma of the same activations, bit for bit.  Reported beside the dense GEMM on the materialised A;
    not `value` (the headline is defined on the reference's (m, n, k) operands, datasets/get_shapes.py:30-40,66-73)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_shapes", os.path.join(ROOT, "datasets", "gen_shapes.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    fused_groups, run_group, spread, ForkedItems = grouping[:4]
    tdt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    N, size, geo = 32, 224, []
    for (cin, cout, ksz, stride, pad) in gen.resnet_convs("resnet50"):   # the reference's walk: sizes chained conv to conv
        osz = gen.conv_out(size, ksz, stride, pad)
        geo.append((cin, cout, ksz, stride, pad, size, osz))
        size = osz
    by_li = sorted(layers, key=lambda L: L["li"])
    convs, ones = [], []
    for L, (cin, cout, ksz, stride, pad, hin, hout) in zip(by_li, geo):
        assert (L["m"], L["n"], L["k"]) == (hout * hout, cout, cin * ksz * ksz) and L["b"] == N, "table row / architecture mismatch"
        if ksz == 1:
            ones.append(L)
            continue
        X = torch.empty(N * cin * hin * hin, dtype=tdt, device=dev)
        sm.fill_uniform(X, 0xC0 + L["li"], 0.0, 1.0)
        need = sm.conv_spmma_workspace(N, cin, hin, hin, ksz, ksz, stride, pad, 1)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        convs.append(dict(L=L, X=X, ws=ws, need=need, g=(N, cin, hin, hin, ksz, ksz, stride, pad, 1), C=torch.empty_like(L["C"])))

    def run_conv(cv):
        sm.conv_spmma(cv["X"], cv["L"]["B"], cv["C"], *cv["g"], cv["L"]["n"], workspace=cv["ws"] if cv["need"] else None)
    items = [("group", Ls) for _, Ls in fused_groups(ones)] + [("single", [dict(cv["L"], conv=cv)]) for cv in convs]
    step = ForkedItems(spread(items), run_group, lambda L: run_conv(L["conv"]))
    t = event_seconds(make_runner(step), max(5, args.steps))
    # verification, after the timed loop: every convolution layer against the pair from the same activations
    ok, checked = True, 0
    for cv in convs:
        L = cv["L"]
        blob = torch.empty(sm.compress24_size(L["m"], L["k"], 2, N), dtype=torch.uint8, device=dev)
        sm.im2col(cv["X"], *cv["g"], blob, compress=True)
        Cref = torch.empty_like(cv["C"])
        sm.spmma(blob, L["B"], Cref, L["m"], L["n"], L["k"], N, 0)
        ok = ok and bool(torch.equal(Cref.view(torch.int16), cv["C"].view(torch.int16)))
        checked += 1
        del blob, Cref
    act_bytes = sum(cv["X"].numel() * 2 for cv in convs)
    a_bytes = sum(cv["L"]["A"].numel() * 2 for cv in convs)
    routes = {}
    for cv in convs:
        r = "im2col-to-blob + matmul" if cv["need"] else "implicit GEMM"
        routes[r] = routes.get(r, 0) + 1
    dense = stages.get("dense_gemm_rowmajor_grouped_ms") or stages.get("dense_gemm_rowmajor_ms")
    res = {"conv_step_ms": t * 1e3, "layers": 49, "conv_layers_from_activations": len(convs), "pointwise_layers_fused_on_nhwc": len(ones), "routes": routes,
           "verified_bit_identical_to_im2col_compress24_plus_spmma": ok, "verified_layers": checked,
           "activation_bytes_instead_of_A_bytes": [act_bytes, a_bytes],
           "dense_gemm_on_materialised_A_ms": dense, "speedup_vs_dense_gemm_on_materialised_A": (dense / (t * 1e3)) if dense else None,
           "headline_step_ms": stages.get("timed_path_ms"),
           "note": "not `value`: the headline stays on the materialised-A configuration the reference's tables define"}
    if not ok:
        sys.stderr.write("bench: conv_step differs from im2col_compress24 + spmma\n")
        raise SystemExit(4)
    return res


def config5_stage(sm, torch, dev):
    """BASELINE config 5: 90 %-sparse COO (one A, density 0.1, values U(-1,1)) x dense, fp32, on four ResNet-50 shapes at
    b = 32 through sm_spmm_coo_f32_ws; HBM GB/s of the algorithmic bytes (B read once + C written once + A) vs the peak."""
    import ctypes
    L_ = sm.lib()
    rows = []
    g = torch.Generator(device=dev).manual_seed(5)
    seen = []
    for sh in read_shapes(table_path("resnet50")):   # every unique shape of the table (four of 17 were timed)
        if sh not in seen:
            seen.append(sh)
    for (m, n, k, b) in seen:
        dense = torch.rand(m, k, generator=g, device=dev) < 0.1
        idx = dense.nonzero()            # row-major scan: sorted by row, then column
        r, c = idx[:, 0].to(torch.int32).contiguous(), idx[:, 1].to(torch.int32).contiguous()
        nnz = int(r.numel())
        v = (torch.rand(nnz, generator=g, device=dev) * 2 - 1).float()
        B = torch.empty(b * k * n, dtype=torch.float32, device=dev)
        sm.fill_uniform(B, 55 + n, -1.0, 1.0)
        C = torch.empty(b * m * n, dtype=torch.float32, device=dev)
        nb = ctypes.c_size_t(0)
        L_.sm_spmm_coo_workspace_size(m, ctypes.byref(nb))
        ws = torch.zeros(nb.value, dtype=torch.uint8, device=dev)
        nb2 = ctypes.c_size_t(0)
        L_.sm_spmm_coo_packed_workspace_size(m, nnz, ctypes.byref(nb2))
        ws2 = torch.zeros(nb2.value, dtype=torch.uint8, device=dev)

        def call_rowptr():
            rc = L_.sm_spmm_coo_f32_ws(m, k, nnz, n, b, r.data_ptr(), c.data_ptr(), v.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                       ws.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(L_.sm_last_error().decode())

        def call_packed():
            rc = L_.sm_spmm_coo_f32_packed(m, k, nnz, n, b, r.data_ptr(), c.data_ptr(), v.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                            ws2.data_ptr(), nb2.value, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(L_.sm_last_error().decode())
        nb3 = ctypes.c_size_t(0)
        L_.sm_spmm_coo_fast_workspace_size(m, k, n, b, ctypes.byref(nb3))
        ws3 = torch.zeros(nb3.value, dtype=torch.uint8, device=dev)

        def call_fast():
            rc = L_.sm_spmm_coo_f32_fast(m, k, nnz, n, b, r.data_ptr(), c.data_ptr(), v.data_ptr(), B.data_ptr(), C.data_ptr(), 1.0, 0.0,
                                          ws3.data_ptr(), nb3.value, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(L_.sm_last_error().decode())
        ms_rowptr = sm.graph_time_ms(call_rowptr, iters=5)
        ms = sm.graph_time_ms(call_packed, iters=5)
        by = nnz * 8.0 + (m + 1) * 4.0 + 4.0 * b * (k * n + m * n)
        row = {"m": m, "n": n, "k": k, "b": b, "nnz": nnz, "ms_exact": ms, "GBs_exact": by / ms / 1e6, "frac_exact": by / ms / 1e6 / HBM_PEAK_GBS,
               "TFs_exact": 2.0 * nnz * n * b / ms / 1e9, "ms_exact_rowptr_form": ms_rowptr}
        form = L_.sm_spmm_coo_fast_form(m, k, nnz, n, b, 0.0) if hasattr(L_, "sm_spmm_coo_fast_form") else (1 if k % 64 == 0 else 0)
        if form:
            ms_fast = sm.graph_time_ms(call_fast, iters=5)
            flag = ctypes.c_int(-1)
            L_.sm_spmm_coo_fast_flag(ws3.data_ptr(), ctypes.byref(flag), None)
            row.update({"ms": ms_fast, "GBs": by / ms_fast / 1e6, "frac": by / ms_fast / 1e6 / HBM_PEAK_GBS, "range_flag": flag.value,
                        "form": ("sparse matrix instruction" if form == 2 else "dense-MFMA") + " (opt-in: strided_coo_options().fast)"})
        else:  # the stem layer's k = 147: the dense-MFMA form takes whole 64-deep stages only; strided_coo runs the exact form
            row.update({"ms": ms, "GBs": by / ms / 1e6, "frac": by / ms / 1e6 / HBM_PEAK_GBS, "form": "exact (packed)", "range_flag": None})
        rows.append(row)
        del B, C, ws, ws2, ws3
    return {"kernel": "ms / GBs / frac = the OPT-IN fast form of sparsifyme::batched::strided_coo (strided_coo_options().fast; the default is the exact fp32 form, ms_exact): sm_spmm_coo_f32_fast -- round 5, `form` = sparse matrix instruction: "
                      "spmm_coo_smfmac_kernel (a 2:4 image of A, hi + lo fp16 planes, + its few third / fourth non-zeros per strip as fp32 entries; B converted in the loader; v_smfmac_f32_16x16x64_f16) after scan / scatter / image kernels, "
                      "whole call timed; `form` = dense-MFMA: the round-4 pipeline (dense operand and A scaled by powers of two computed on the "
                      "device, rounded to fp16 / split hi + lo, fp16 MFMA with fp32 accumulation, inverse scales on the fp32 sums; result within 2^-11 of "
                      "sum|a||b| at any magnitude; a range flag + untouched C when an operand does not convert -> exact fallback; whole call incl. its scan / "
                      "conversion / scatter passes) = ms / GBs / frac; the exact forms beside it: ms_exact = sm_spmm_coo_f32_packed (re-ordering of A + product), "
                      "ms_exact_rowptr_form = sm_spmm_coo_f32_ws", "shapes": rows,
            "unit": "GB/s of algorithmic bytes (SURVEY.md 8(d): nnz*(s+4) + (m+1)*4 + b*s*(k*n + m*n))", "peak": HBM_PEAK_GBS}


def bell_stage(sm, torch, dev):
    """batched::spmm on Blocked-ELL operands (the reference's only recorded sparse number, examples/compare.csv column
    `spmm`; call spmm.hxx:94-111) as examples/spmm.cu builds them: 2 x 2 blocks, half of the block columns present, one A per
    batch index, B shared, fp32, b = 32, all batches in one submission (sm_spmm_bell_batched_f32).  Bytes = stored values +
    block indices + B + C; the kernel pair expands the blocks and runs the dense fp32 MFMA product, so the binding roofline is
    max(bytes / HBM peak, dense flops / fp32 matrix peak).  Host pointer tables: not graph-capturable, wall clock over 5 calls."""
    import ctypes
    L_ = sm.lib()
    rows = []
    g = torch.Generator(device=dev).manual_seed(11)
    for (m, n, k, b) in [(784, 256, 2304, 32), (12544, 64, 576, 32), (196, 512, 4608, 32), (3136, 128, 1152, 32)]:
        bs, ell_cols = 2, k // 2
        bcols = ell_cols // bs
        vals, idxs = [], []
        for _ in range(b):
            ci = torch.rand(m // bs, k // bs, generator=g, device=dev).argsort(dim=1)[:, :bcols].sort(dim=1).values
            idxs.append(ci.to(torch.int64).contiguous().view(-1))
            v = torch.empty(m * ell_cols, dtype=torch.float32, device=dev)
            sm.fill_uniform(v, 77 + len(vals), -0.5, 0.5)
            vals.append(v)
        B = torch.empty(k * n, dtype=torch.float32, device=dev)
        sm.fill_uniform(B, 78, -0.5, 0.5)
        Cs = [torch.empty(m * n, dtype=torch.float32, device=dev) for _ in range(b)]
        nb = ctypes.c_size_t(0)
        L_.sm_spmm_bell_batched_workspace_size(m, k, b, ctypes.byref(nb))
        ws = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        PA = ctypes.c_void_p * b
        pv, pi, pc = PA(*[v.data_ptr() for v in vals]), PA(*[i.data_ptr() for i in idxs]), PA(*[c.data_ptr() for c in Cs])

        def call():
            rc = L_.sm_spmm_bell_batched_f32(pv, pi, m, k, bs, ell_cols, B.data_ptr(), pc, n, b, 1.0, 0.0, ws.data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                raise RuntimeError(L_.sm_last_error().decode())
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # device time by a HIP event pair on the launch stream (round 3 timed this stage by wall clock)
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        by = b * (m * ell_cols * 4.0 + (m // bs) * bcols * 8.0 + m * n * 4.0) + k * n * 4.0
        fl = 2.0 * m * n * k * b  # what the expanded fp32 MFMA product executes (stored-value flops are half of it)
        roof_ms = max(by / (HBM_PEAK_GBS * 1e9), fl / (F32_MATRIX_PEAK_TFS * 1e12)) * 1e3
        rows.append({"m": m, "n": n, "k": k, "b": b, "block": bs, "ell_cols": ell_cols, "ms": ms, "bytes": by, "GBs": by / ms / 1e6,
                     "executed_TFs": fl / ms / 1e9, "stored_value_TFs": fl / 2 / ms / 1e9,
                     "bound": "mfma" if fl / (F32_MATRIX_PEAK_TFS * 1e12) > by / (HBM_PEAK_GBS * 1e9) else "hbm", "frac": roof_ms / ms})
        del vals, idxs, Cs, ws
    return {"kernel": "bell_expand_rows_kernel + gemm_f32_dma_kernel (sm_spmm_bell_batched_f32, one submission for all batches)",
            "shapes": rows, "frac": "roofline time / measured time, roofline = max(algorithmic bytes / 8 TB/s, executed dense flops / 157.3 TF/s)",
            "timing": "HIP event pair on the launch stream around 5 calls after one warm-up (host pointer tables: the entry point is not graph-capturable)"}
