#!/usr/bin/env python3
"""bench.py -- the hot path on BASELINE.json's headline workload.

Metric: effective GF/s (2:4 spmma vs dense gemm) on the ResNet-50 layer shapes, fp16, b = 32.
A "step" is one pass of the hot path over the whole table: for each of the 49 layers the 2:4 prune + compress of the
per-batch activation operand A and the sparse x dense matmul -- as ONE fused kernel (sm_spmma_fused_f16) on the layers
where that wins (--path auto: n <= 256 or k <= 512, 42 layers) and as sm_compress24_f16 + sm_spmma_f16 on the rest; both
give the same C bit for bit.  value = dense-equivalent flops (2*m*n*k*b summed over the table, summed over ranks) / time.
Inputs are generated on the device and are resident in HBM before the timed region starts.  The step's 56 launches,
spread over 4 HIP streams, are captured once into a hipGraph and replayed (the launches are 7-100 us each; without the
graph the Python/ctypes call cost would be on the clock).

Multi-GPU: one process per GPU; every rank runs the same table on its own seeded batch (weak
scaling, no data-path collective); one tiny all-reduce (RCCL) gives sum(flops) and max(time).

Besides the contract line's fields the JSON carries `stages` (matmul only, compress only, and the
dense GEMMs that are the metric's denominator), `roofline` for the dominant kernel of the timed step
and `cpu_baseline` (the oracle's arithmetic timed on the host cores; rank 0, N = 1 only).
"""
import argparse
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def read_shapes(path):
    with open(path, newline="") as f:
        rows = list(csv.reader(f))[1:]
    return [tuple(int(x) for x in r[:4]) for r in rows if r]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--table", default=os.path.join(ROOT, "datasets", "resnet50.csv"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-stage / denominator passes")
    ap.add_argument("--eager", action="store_true", help="launch from Python instead of replaying a hipGraph")
    ap.add_argument("--path", choices=["auto", "staged"], default="auto",
                    help="auto: fused prune+compress+matmul kernel on the layers where it wins (n <= --fused-max-n), the "
                         "staged compress24 + spmma pair elsewhere; staged: the pair on every layer")
    ap.add_argument("--fused-max-n", type=int, default=256,
                    help="auto path: widest n served by sm_spmma_fused_f16 whatever k (one workgroup spans up to 256 columns, so up "
                         "to there A is loaded and selected once).  512 also fuses the long-k n = 512 layers, which standalone are "
                         "faster as compress + spmma (46 vs 70 us) but inside the step, bound by the bytes it moves, gain from not "
                         "writing and re-reading their blobs: 1.816 vs 1.837 ms (three alternating runs) -- not the default because "
                         "it makes the dominant kernel family's single-stream roofline figure worse (0.41 vs 0.47)")
    ap.add_argument("--fused-max-k-wide", type=int, default=512,
                    help="auto path: wider layers (n > --fused-max-n) are still fused when k <= this (the A-stationary "
                         "kernel keeps the 2:4 image of a row panel in LDS across its column tiles); 0 = never")
    ap.add_argument("--dtype", choices=["f16", "bf16"], default="f16",
                    help="element type of the 16-bit path (BASELINE's metric is quoted on f16; bf16 runs the same kernels "
                         "with the bfloat16 matrix instructions; the reference's column-major dense GEMM has no bf16 form "
                         "and is skipped)")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the independent layers of a step are spread over (fork/join inside the step)")
    ap.add_argument("--graphs", choices=["single", "per-stream"], default="single",
                    help="hipGraph form of a step: one graph holding every chain (default) or one linear graph per stream")
    ap.add_argument("--sched", choices=["rr", "split"], default="rr",
                    help="layer -> stream assignment: round-robin, or chip-filling layers (>= 784 row tiles) on the first half "
                         "of the streams and the under-filling ones on the second half")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="multi-rank rehearsal on a ONE-GPU box: gloo backend, every rank on cuda:0 (control flow only; "
                         "the ranks share the device, so the numbers mean nothing)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_gloo:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    sm = ge.load_package()
    mg = ge.load_package_module("multigpu")
    sm.device_check()  # raises when the HIP library or a gfx950 device is missing: no fallback

    shapes = read_shapes(args.table)
    tdt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    layers = []
    for li, (m, n, k, b) in enumerate(shapes):
        A = torch.empty(b * m * k, dtype=tdt, device=dev)
        B = torch.empty(k * n, dtype=tdt, device=dev)
        sm.fill_uniform(A, 0x5EED0000 + 1000 * rank + li, 0.0, 1.0)
        sm.fill_uniform(B, 0xB0000000 + li, 0.0, 1.0)
        blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
        C = torch.empty(b * m * n, dtype=tdt, device=dev)
        layers.append(dict(m=m, n=n, k=k, b=b, A=A, B=B, blob=blob, C=C))
    flops = sum(2.0 * L["m"] * L["n"] * L["k"] * L["b"] for L in layers)

    # The 49 layers of a step are independent problems (the reference's sweep runs them as separate
    # processes, examples/profiling.py:6-17), so a step forks them over a few HIP streams and joins:
    # one layer's ramp-up and tail overlap another layer's streaming phase.  compress -> spmma of one
    # layer stay ordered on one stream.
    side = [torch.cuda.Stream() for _ in range(max(0, args.streams - 1))]

    nstreams = len(side) + 1
    chains = [[] for _ in range(nstreams)]  # stream w runs chains[w] in order
    cnt = [0, 0]
    for li, L in enumerate(layers):
        w = li % nstreams
        if args.sched == "split" and nstreams >= 2:
            big = 0 if L["m"] * L["b"] >= 784 * 128 else 1
            half = [nstreams // 2, nstreams - nstreams // 2]
            w = (0 if big == 0 else half[0]) + cnt[big] % half[big]
            cnt[big] += 1
        chains[w].append(L)

    class Forked:
        """A step whose layers are spread over the streams: fork, one chain of layers per stream, join."""

        def __init__(self, per_layer):
            self.per_layer = per_layer

        def fork_join(self, run_chain):
            main = torch.cuda.current_stream()
            for s_ in side:
                s_.wait_stream(main)
            run_chain(0)
            for w, s_ in enumerate(side, start=1):
                with torch.cuda.stream(s_):
                    run_chain(w)
            for s_ in side:
                main.wait_stream(s_)

        def chain(self, w):
            for L in chains[w]:
                self.per_layer(L)

        def __call__(self):  # launched kernel by kernel
            self.fork_join(self.chain)

    def layer_full(L):
        sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"])
        sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)

    def use_fused(L):
        return args.path == "auto" and L["k"] % 64 == 0 and (L["n"] <= args.fused_max_n or L["k"] <= args.fused_max_k_wide)

    # (f-1) the fused kernel computes the same C bit for bit straight from the dense A (the 2:4 selection
    # and compaction happen in registers / LDS; no blob goes to HBM)
    def layer_path(L):
        if use_fused(L):
            sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"])
        else:
            layer_full(L)

    step_full = Forked(layer_path)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_runner(fn):
        """fn replayed from a hipGraph (one graph = one call of fn), or fn itself with --eager.  --graphs per-stream
        turns a Forked step into one linear graph per stream, replayed into its own stream between the fork and the
        join.  Measured the same as the single graph (1.87 vs 1.88 ms): under either form the hardware runs two or
        three of the four chains at a time (profiles/ktrace_r01l.txt, tools/ktrace_step.py), with eager launches all
        four, and the step takes the same time in all three cases -- it is bound by the HBM rate of the kernel mix,
        not by how the chains interleave."""
        fn()  # first call outside capture: lazy module loads, function attributes
        torch.cuda.synchronize()
        if args.eager:
            return fn
        try:
            if isinstance(fn, Forked) and args.graphs == "per-stream":
                graphs = []
                for w in range(nstreams):
                    gw = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gw, stream=torch.cuda.Stream()):
                        fn.chain(w)
                    graphs.append(gw)
                return lambda: fn.fork_join(lambda w: graphs[w].replay())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=torch.cuda.Stream()):
                fn()
            return g.replay
        except Exception as e:  # capture unsupported: fall back to eager launches, and say so
            sys.stderr.write(f"bench: hipGraph capture failed ({e}); launching eagerly\n")
            return fn

    def timed(run, steps, warmup, collective=True):
        """collective=False: rank-local timing (the per-stage / per-family passes only rank 0 runs: a distributed
        barrier there would pair with the other ranks' final barrier and hang the job)."""
        sync = barrier if collective else torch.cuda.synchronize
        for _ in range(warmup):
            run()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        sync()
        return wall

    run_full = make_runner(step_full)
    wall = timed(run_full, args.steps, args.warmup)
    tot_flops, wall_max = mg.rollup(flops * args.steps, wall, None if args.rehearse_gloo else dev)
    ms_per_step = wall_max / args.steps * 1e3
    value = tot_flops / wall_max / 1e9

    out = {
        "metric": "effective GF/s (2:4 spmma vs dense gemm) on ResNet-50 layer shapes",
        "value": value, "unit": "GF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "datasets/%s: %d conv layers as im2col GEMMs (m,n,k) at b=%d, %s; "
                               "step = per layer 2:4 prune+compress+matmul (path: %s)"
                               % (os.path.basename(args.table), len(layers), layers[0]["b"], "fp16" if args.dtype == "f16" else "bfloat16", args.path),
                   "path": args.path + (": sm_spmma_fused_f16 on %d layers (n <= %d or k <= %d), sm_compress24_f16 + sm_spmma_f16 on %d"
                                        % (sum(use_fused(L) for L in layers), args.fused_max_n, args.fused_max_k_wide,
                                           sum(not use_fused(L) for L in layers))
                                        if args.path == "auto" else ": sm_compress24_f16 + sm_spmma_f16 on every layer"),
                   "layers": len(layers), "batch": layers[0]["b"], "dense_equiv_gflop_per_step": flops / 1e9,
                   "launch": "eager" if args.eager else ("hipGraph replay, one linear graph per stream" if args.graphs == "per-stream" else "hipGraph replay of one step"),
                   "streams": args.streams, "sched": args.sched,
                   "parallelism": f"replicated table x{world}, per-rank batch, no data-path collective"},
    }

    if rank == 0 and not args.no_extras:
        R = max(5, args.steps)

        def sec_per_call(fn):
            return timed(make_runner(fn), R, 2, collective=False) / R

        spmma_only = Forked(lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0))
        compress_only = Forked(lambda L: sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]))
        dense_rowmajor = Forked(lambda L: sm.gemm_rowmajor(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]))

        # the reference's dense path: column-major pointer-array batched GEMM, B shared (examples/gemm.cu:60,86)
        for L in layers:
            m, n, k, b = L["m"], L["n"], L["k"], L["b"]
            L["Ap"] = torch.tensor([L["A"].data_ptr() + 2 * i * m * k for i in range(b)], dtype=torch.int64, device=dev)
            L["Bp"] = torch.tensor([L["B"].data_ptr()] * b, dtype=torch.int64, device=dev)
            L["Cp"] = torch.tensor([L["C"].data_ptr() + 2 * i * m * n for i in range(b)], dtype=torch.int64, device=dev)

        dense_batched = Forked(lambda L: sm.gemm_batched(L["Ap"], L["Bp"], L["Cp"], L["m"], L["n"], L["k"], L["b"], "f16"))

        t_mul, t_cmp = sec_per_call(spmma_only), sec_per_call(compress_only)
        t_drm = sec_per_call(dense_rowmajor)
        t_dcm = sec_per_call(dense_batched) if args.dtype == "f16" else None  # cublasHgemmBatched's role: fp16 only
        t_full = wall / args.steps

        step_staged = Forked(layer_full)

        t_staged = t_full if args.path == "staged" else sec_per_call(step_staged)
        gfs = lambda t: flops / t / 1e9
        out["stages"] = {
            "spmma_mul_gfs": gfs(t_mul), "spmma_mul_ms": t_mul * 1e3, "compress_ms": t_cmp * 1e3,
            "dense_gemm_rowmajor_gfs": gfs(t_drm), "dense_gemm_rowmajor_ms": t_drm * 1e3,
            "dense_gemm_batched_colmajor_gfs": gfs(t_dcm) if t_dcm else None, "dense_gemm_batched_colmajor_ms": t_dcm * 1e3 if t_dcm else None,
            "speedup_mul_vs_dense_rowmajor": t_drm / t_mul, "speedup_mul_vs_dense_batched": t_dcm / t_mul if t_dcm else None,
            "speedup_full_vs_dense_rowmajor": t_drm / t_full, "speedup_full_vs_dense_batched": t_dcm / t_full if t_dcm else None,
            "full_path_staged_gfs": gfs(t_staged), "full_path_staged_ms": t_staged * 1e3,
            "timed_path": args.path, "timed_path_ms": t_full * 1e3,
            # what 2:4 can buy on these shapes when both products are HBM-bound (they are: DESIGN.md 4.2): the ratio of
            # the algorithmic bytes, dense (A + B + C) over sparse (9/16 A + B + C)
            "hbm_bound_speedup_ceiling": sum(L["b"] * 2.0 * (L["m"] * L["k"] + L["m"] * L["n"]) + 2.0 * L["k"] * L["n"] for L in layers)
            / sum(L["b"] * (L["m"] * L["k"] * (1.0 + 1.0 / 8) + 2.0 * L["m"] * L["n"]) + 2.0 * L["k"] * L["n"] for L in layers),
        }
        # matrix-pipe view of the 2:4 matmul (north_star: "MFMA utilisation for the matmul against chip peak"):
        # dense-equivalent rate of the matmul-only pass against 2 x the dense fp16 peak (v_smfmac does a 16x16x64
        # product in the cycles of a dense 16x16x32), plus the PMC MfmaUtil per kernel when a profile is present
        mfma = {"achieved_TFs": gfs(t_mul) / 1e3, "peak_TFs": 2.0 * 2500.0, "frac": gfs(t_mul) / 1e3 / 5000.0,
                "peak": "2 x 2.5 PF/s dense fp16 (MI355X_MICROARCH.md); the v_smfmac issue rate measured on this chip "
                        "is 3.4-3.8 PF/s dense-equivalent (profiles/mfma_rate_r01.txt)",
                "pmc_mfma_util_percent": None}
        mpath = os.path.join(ROOT, "profiles", "mfma_util_latest.json")  # tools/pmc_mfma.py, from a rocprofv3 --pmc pass
        if os.path.exists(mpath):
            try:
                mfma["pmc_mfma_util_percent"] = {k: round(v["mfma_util_percent"], 2) for k, v in json.load(open(mpath)).items()}
            except Exception:
                pass
        out["stages"]["matmul_mfma"] = mfma
        # roofline of the dominant kernel family of the timed step: algorithmic bytes (SURVEY.md 8(d),
        # DESIGN.md 4) / time of a single-stream pass that launches only that family on the layers it serves
        s = 2
        fam = {"spmma_f16": dict(names=["spmma_f16_dma_kernel", "spmma_f16_pc_kernel", "spmma_f16_kernel"],
                                 layers=[L for L in layers if not use_fused(L)],
                                 call=lambda L: sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0),
                                 bytes=lambda L: L["b"] * (L["m"] * L["k"] * s / 2 + L["m"] * L["k"] / 8 + L["m"] * L["n"] * s) + s * L["k"] * L["n"]),
               "compress": dict(names=["compress_flat_kernel", "compress_rowspan_f16_kernel", "compress_kernel"],
                                layers=[L for L in layers if not use_fused(L)],
                                call=lambda L: sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"]),
                                bytes=lambda L: L["b"] * L["m"] * L["k"] * (s + s / 2 + 1 / 8)),
               "spmma_f16_fused": dict(names=["spmma_f16_fused_direct_kernel", "spmma_f16_fused_wide_kernel", "spmma_f16_fused_astat_kernel"],
                                       layers=[L for L in layers if use_fused(L)],
                                       call=lambda L: sm.spmma_fused(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"]),
                                       bytes=lambda L: L["b"] * s * (L["m"] * L["k"] + L["m"] * L["n"]) + s * L["k"] * L["n"])}
        traffic_tab = {}
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")  # tools/pmc_traffic.py, from rocprofv3 --pmc passes
        if os.path.exists(tpath):
            try:
                traffic_tab = json.load(open(tpath))
            except Exception:
                traffic_tab = {}
        rows = {}
        for name, f in fam.items():
            if not f["layers"]:
                continue

            def serial(f=f):
                for L in f["layers"]:
                    f["call"](L)
            t = sec_per_call(serial)
            by = sum(f["bytes"](L) for L in f["layers"])
            tb = [(traffic_tab[n]["hbm_bytes_per_launch"], traffic_tab[n]["launches_profiled"]) for n in f["names"] if n in traffic_tab]
            traffic = sum(b_ * c_ for b_, c_ in tb) / sum(c_ for _, c_ in tb) if tb else None
            rows[name] = dict(seconds=t, launches=len(f["layers"]), bytes=by, GBs=by / t / 1e9, traffic=traffic)
        dom = max(rows, key=lambda n_: rows[n_]["seconds"])
        d = rows[dom]
        out["roofline"] = {"bound": "hbm", "achieved": d["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": d["GBs"] / HBM_PEAK_GBS, "traffic": d["traffic"], "kernel": dom,
                           "launches_per_step": d["launches"], "avg_launch_us": d["seconds"] / d["launches"] * 1e6,
                           "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                           "measured": "single stream, one kernel family at a time, hipGraph replay",
                           "families": {n_: {"ms_per_step": r_["seconds"] * 1e3, "launches": r_["launches"], "GBs": r_["GBs"],
                                             "hbm_traffic_per_launch": r_["traffic"]} for n_, r_ in rows.items()}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ge, shapes)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def cpu_baseline(ge, shapes):
    """The oracle's arithmetic ('port': fp32 accumulate, OpenMP over rows) on the host cores, on a
    bounded sample: one batch (b = 1) of every unique (m,n,k) of the table, repeated; both the dense
    product and the 2:4 path (STRIP selection fused with the two kept MACs per strip)."""
    import numpy as np
    orc = ge.load_oracle()
    uniq = sorted(set((m, n, k) for m, n, k, _ in shapes))
    rng = np.random.default_rng(0x5EED)
    reps = 64
    fl = t_dense = t_sparse = 0.0
    for (m, n, k) in uniq:
        r = m
        A = rng.uniform(0, 1, r * k).astype(np.float32)
        B = rng.uniform(0, 1, k * n).astype(np.float32)
        C = np.zeros(r * n, dtype=np.float32)
        orc.cpu_gemm_f32(A, B, C, r, n, k)  # warm
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.cpu_gemm_f32(A, B, C, r, n, k)
        t_dense += time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.cpu_spmma_f32(A, B, C, r, n, k)
        t_sparse += time.perf_counter() - t0
        fl += 2.0 * r * n * k * reps
    return {"value": fl / t_sparse / 1e9, "unit": "GF/s", "cores": orc.num_threads(), "kind": "port",
            "dense_value": fl / t_dense / 1e9,
            "sample": f"oracle sm_cpu_spmma_f32 (2:4 path) / sm_cpu_gemm_f32 (dense_value), fp32, one batch (b=1) of each of "
                      f"the {len(uniq)} unique ResNet-50 shapes x {reps} repetitions ({fl / 1e9:.1f} dense-equivalent GFLOP, "
                      f"{t_dense + t_sparse:.1f} s of CPU work); effective GF/s = dense-equivalent flops / time"}


if __name__ == "__main__":
    main()
