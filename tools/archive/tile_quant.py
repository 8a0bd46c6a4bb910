#!/usr/bin/env python3
"""Tile-quantisation probe: time sm_spmma_fused_f16 on row counts around multiples of the workgroup slots
(256 CUs x resident workgroups) to see what the last partial round of tiles costs.  tools/tile_quant.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
cases = [(128, 1152, [2048, 3072, 3136, 3200, 4096]), (128, 512, [3072, 3136]), (256, 512, [2048, 3072, 3136, 4096]),
         (256, 2304, [512, 768, 784, 1024]), (256, 1024, [768, 784, 1024]), (64, 576, [12288, 12544])]
b = 32
for n, k, ms_list in cases:
    for m in ms_list:
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
        tiles = (m * b + 127) // 128
        by = b * m * (k + n) * 2 + k * n * 2
        print(f"n={n:4d} k={k:5d} m={m:6d} tiles={tiles:5d} ({tiles / 256:6.2f} per CU)  {t:8.1f} us  {by / t / 1e3:7.0f} GB/s  {t / tiles * 1e3:7.1f} ns/tile", flush=True)
