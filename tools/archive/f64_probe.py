#!/usr/bin/env python3
"""Device time of sm_gemm_batched_f64 (column-major pointer-array batched, the reference's cublasDgemmBatched call) on a
few shapes.  tools/f64_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
for (m, n, k, b) in [(2048, 2048, 2048, 1), (4096, 4096, 1024, 1), (3136, 128, 1152, 8), (784, 256, 2304, 8), (12544, 64, 576, 4)]:
    As = [torch.rand(m * k, dtype=torch.float64, device=dev) for _ in range(b)]
    B = torch.rand(k * n, dtype=torch.float64, device=dev)
    Cs = [torch.zeros(m * n, dtype=torch.float64, device=dev) for _ in range(b)]
    ptr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)
    Ap, Bp, Cp = ptr(As), ptr([B] * b), ptr(Cs)
    f = lambda: sm.gemm_batched(Ap, Bp, Cp, m, n, k, b, "f64")
    t = sm.graph_time_ms(f, iters=3, replays=2)
    print(f"{m}x{n}x{k} b={b}: {t:8.3f} ms  {2.0 * m * n * k * b / t / 1e9:7.2f} TF/s", flush=True)
