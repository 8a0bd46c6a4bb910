#!/usr/bin/env python3
"""bench.py -- the hot path on BASELINE.json's headline workload.

Metric: effective GF/s (2:4 spmma vs dense gemm) on the ResNet-50 layer shapes, fp16, b = 32.
A "step" is one pass of the hot path over the whole table: for each of the 49 layers, compress
(the fused 2:4 prune + compress of the per-batch activation operand A) and then the 2:4
sparse x dense matmul.  value = dense-equivalent flops (2*m*n*k*b summed over the table, times the
number of ranks) / time.  Inputs are generated on the device and are resident in HBM before the
timed region starts.  Multi-GPU: every rank runs the same table on its own seeded batch (weak
scaling, no data-path collective); one tiny all-reduce (RCCL) takes the max time over ranks.

Besides the contract line's fields the JSON carries: per-stage throughputs (matmul only, compress,
prune, the dense GEMMs that are the metric's denominator), `roofline` for the dominant kernel and
`cpu_baseline` (the oracle's arithmetic timed on the host cores; rank 0, N = 1 only).
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
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--table", default=os.path.join(ROOT, "datasets", "resnet50.csv"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-stage / denominator passes")
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
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    sm = ge.load_package()
    sm.device_check()  # raises when the HIP library or a gfx950 device is missing: no fallback

    shapes = read_shapes(args.table)
    layers = []
    for li, (m, n, k, b) in enumerate(shapes):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev)
        B = torch.empty(k * n, dtype=torch.float16, device=dev)
        sm.fill_uniform(A, 0x5EED0000 + 1000 * rank + li, 0.0, 1.0)
        sm.fill_uniform(B, 0xB0000000 + li, 0.0, 1.0)
        blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
        C = torch.empty(b * m * n, dtype=torch.float16, device=dev)
        layers.append(dict(m=m, n=n, k=k, b=b, A=A, B=B, blob=blob, C=C))
    flops = sum(2.0 * L["m"] * L["n"] * L["k"] * L["b"] for L in layers)

    def step_full():
        for L in layers:
            sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"])
            sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        barrier()
        return wall, e0.elapsed_time(e1) * 1e-3

    wall, _ = timed(step_full, args.steps, args.warmup)
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall / args.steps * 1e3
    value = flops * world / (wall / args.steps) / 1e9

    out = {
        "metric": "effective GF/s (2:4 spmma vs dense gemm) on ResNet-50 layer shapes",
        "value": value, "unit": "GF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "datasets/resnet50.csv: 49 conv layers as im2col GEMMs (m,n,k) at b=32, fp16; "
                               "step = per layer compress24 (fused 2:4 prune+compress of A) + 2:4 spmma",
                   "layers": len(layers), "batch": layers[0]["b"], "dense_equiv_gflop_per_step": flops / 1e9,
                   "parallelism": f"replicated table x{world}, per-rank batch, no data-path collective"},
    }

    if rank == 0 and not args.no_extras:
        R = max(3, args.steps)

        def dev_time(fn):
            _, t = timed(fn, R, 1)
            return t / R

        def spmma_only():
            for L in layers:
                sm.spmma(L["blob"], L["B"], L["C"], L["m"], L["n"], L["k"], L["b"], 0)

        def compress_only():
            for L in layers:
                sm.compress24(L["A"], L["m"], L["k"], L["k"], L["b"], L["m"] * L["k"], L["blob"])

        def dense_rowmajor():
            for L in layers:
                sm.gemm_rowmajor(L["A"], L["B"], L["C"], L["m"], L["n"], L["k"], batch=L["b"])

        # the reference's dense path: column-major pointer-array batched GEMM, B shared (examples/gemm.cu:60,86)
        for L in layers:
            m, n, k, b = L["m"], L["n"], L["k"], L["b"]
            L["Ap"] = torch.tensor([L["A"].data_ptr() + 2 * i * m * k for i in range(b)], dtype=torch.int64, device=dev)
            L["Bp"] = torch.tensor([L["B"].data_ptr()] * b, dtype=torch.int64, device=dev)
            L["Cp"] = torch.tensor([L["C"].data_ptr() + 2 * i * m * n for i in range(b)], dtype=torch.int64, device=dev)

        def dense_batched():
            for L in layers:
                sm.gemm_batched(L["Ap"], L["Bp"], L["Cp"], L["m"], L["n"], L["k"], L["b"], "f16")

        t_mul, t_cmp = dev_time(spmma_only), dev_time(compress_only)
        t_drm, t_dcm = dev_time(dense_rowmajor), dev_time(dense_batched)
        t_full = dev_time(step_full)
        gfs = lambda t: flops / t / 1e9
        out["stages"] = {
            "spmma_mul_gfs": gfs(t_mul), "compress_ms": t_cmp * 1e3, "spmma_mul_ms": t_mul * 1e3,
            "full_path_device_ms": t_full * 1e3,
            "dense_gemm_batched_colmajor_gfs": gfs(t_dcm), "dense_gemm_rowmajor_gfs": gfs(t_drm),
            "speedup_mul_vs_dense_batched": t_dcm / t_mul, "speedup_mul_vs_dense_rowmajor": t_drm / t_mul,
            "speedup_full_vs_dense_batched": t_dcm / t_full,
        }
        # roofline of the dominant kernel of the timed step
        s = 2
        by_spmma = sum(L["b"] * (L["m"] * L["k"] * s / 2 + L["m"] * L["k"] / 8 + L["m"] * L["n"] * s) + s * L["k"] * L["n"]
                       for L in layers)
        by_cmp = sum(L["b"] * L["m"] * L["k"] * (s + s / 2 + 1 / 8) for L in layers)
        dom, by, t = ("spmma_f16_kernel", by_spmma, t_mul) if t_mul >= t_cmp else ("compress_kernel", by_cmp, t_cmp)
        ach = by / t / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": None, "kernel": dom,
                           "launches_per_step": len(layers), "avg_launch_us": t / len(layers) * 1e6,
                           "algorithmic_bytes_per_step": by,
                           "other": {"compress_kernel_GBs": by_cmp / t_cmp / 1e9, "spmma_f16_kernel_GBs": by_spmma / t_mul / 1e9}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(ge, shapes)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def cpu_baseline(ge, shapes):
    """The oracle's arithmetic ('port': fp32 accumulate, OpenMP over rows) on the host cores, on a
    bounded sample: for every unique (m,n,k) of the table, `rows` rows of one batch; both the dense
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
