#!/usr/bin/env python3
"""Times sm_spmma_fused_f16 on the wide short-K layers of the ResNet-50 table (A/B runs under env switches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
b = 32
for (m, n, k) in [(3136, 512, 128), (784, 1024, 256), (196, 2048, 512), (12544, 256, 64)]:
    sets = []
    for i in range(3):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, 0.0, 1.0)
        C = torch.empty(b * m * n, dtype=torch.float16, device=dev)
        sets.append((A, C))
    B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 2, 0.0, 1.0)
    it = [0]
    def f():
        A, C = sets[it[0] % 3]; it[0] += 1
        sm.spmma_fused(A, B, C, m, n, k, batch=b)
    t = sm.graph_time_ms(f, iters=12, replays=3) * 1e3
    by = b * m * (k + n) * 2 + k * n * 2
    print(f"{m:6d} {n:4d} {k:5d}: {t:8.1f} us  {by / t / 1e3:6.0f} GB/s", flush=True)
