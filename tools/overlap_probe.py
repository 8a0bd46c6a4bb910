#!/usr/bin/env python3
"""How well do a bandwidth-bound launch (direct family) and a few-tile launch (big / wide / A-stationary) overlap when issued on two streams?
Each alone, then both at once (eager, two plain streams; launches are >= 30 us, so the issue cost does not matter).  python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
b = 32
def group(m, n, k, cnt, seed):
    As, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, seed + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, seed + 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    return lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
def t(fns_streams, reps=20):
    for f, s in fns_streams:
        with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for f, s in fns_streams:
            with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
direct = {"12544x64x576 x3": group(12544, 64, 576, 3, 1), "3136x128x1152 x4": group(3136, 128, 1152, 4, 100)}
few = {"196x512x4608 x3 (big)": group(196, 512, 4608, 3, 200), "784x256x2304 x6 (wide)": group(784, 256, 2304, 6, 300), "784x1024x256 x6 (astat)": group(784, 1024, 256, 6, 400),
       "784x256x1024 x5 (big)": group(784, 256, 1024, 5, 500)}
for dn, df in direct.items():
    td = t([(df, s1)])
    for fn, ff in few.items():
        tf = t([(ff, s2)])
        both = t([(df, s1), (ff, s2)])
        both_r = t([(ff, s2), (df, s1)])
        print(f"{dn:18s} {td:6.1f} us | {fn:26s} {tf:6.1f} us | together {both:6.1f} (few-tile first: {both_r:6.1f})  serial {td + tf:6.1f}  overlap saves {100 * (1 - min(both, both_r) / (td + tf)):4.1f} %", flush=True)
