#!/usr/bin/env python3
"""Times sm_spmma_f16 on the small-M long-K layers of the ResNet-50 table (A/B runs under env switches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
b = 32
for (m, n, k) in [(196, 512, 4608), (196, 512, 2048), (784, 512, 1024), (196, 2048, 512)]:
    sets = []
    for i in range(4):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, 0.0, 1.0)
        blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
        sm.compress24(A, m, k, k, b, m * k, blob)
        C = torch.empty(b * m * n, dtype=torch.float16, device=dev)
        sets.append((blob, C))
    B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 2, 0.0, 1.0)
    it = [0]
    def f():
        blob, C = sets[it[0] % 4]; it[0] += 1
        sm.spmma(blob, B, C, m, n, k, b, 0)
    t = sm.graph_time_ms(f, iters=12, replays=3) * 1e3
    print(f"{m:6d} {n:4d} {k:5d}: {t:8.1f} us", flush=True)
