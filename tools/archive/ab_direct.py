#!/usr/bin/env python3
"""Times sm_spmma_fused_f16 on the direct-kernel shapes of the ResNet-50 table (for A/B runs under env switches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
b = 32
tot = 0.0
for (m, n, k, cnt) in [(12544, 64, 64, 1), (12544, 64, 576, 3), (12544, 64, 256, 2), (12544, 128, 256, 1), (12544, 256, 64, 3),
                       (3136, 128, 1152, 4), (3136, 128, 512, 3)]:
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
    tot += t * cnt
    print(f"{m:6d} {n:4d} {k:5d} x{cnt}: {t:8.1f} us  {by / t / 1e3:6.0f} GB/s", flush=True)
print(f"weighted total {tot:.0f} us")
