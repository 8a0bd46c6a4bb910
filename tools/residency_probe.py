#!/usr/bin/env python3
"""Round 6: can a few-tile kernel that LEAVES ROOM on its CU overlap with a direct-family launch?
The tuning library's SM_FUSED_D256=1 runs the n > 128 shapes on 128 x 256 direct tiles (4 waves, 96 KiB of LDS, 300 registers: one such workgroup per
CU and room for one 48-64 KiB direct workgroup of another launch beside it), where the product library runs the big / wide / A-stationary kernels
(whole LDS or 16 waves per workgroup).  For each (direct launch D, few-tile launch F): each alone, both at once on two streams, and the model
t_F + max(0, t_D - t_F * idle_CUs / 256) with idle_CUs = 256 - min(256, tiles of F) -- what overlap by CU count alone would give.
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/residency_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
sm = ge.load_package()
assert "tuning" in sm.LIB_PATH, "run with SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so"
dev = torch.device("cuda", 0)
b = 32


def group(m, n, k, cnt, seed):
    As, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, seed + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, seed + 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    return (lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b)), Cs


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


direct = {"12544x64x576 x3": group(12544, 64, 576, 3, 1)[0], "3136x128x1152 x4": group(3136, 128, 1152, 4, 100)[0]}
few = {"196x512x4608 x3": (196, 512, 4608, 3), "784x256x2304 x6": (784, 256, 2304, 6), "784x256x1024 x5": (784, 256, 1024, 5), "784x512x1024 x1": (784, 512, 1024, 1)}
print("# few-tile launch F: product dispatch (big / wide) vs 128 x 256 direct tiles (SM_FUSED_D256=1: 96 KiB, 4 waves, one per CU + room for a direct workgroup)")
for fn, (m, n, k, cnt) in few.items():
    ff, Cs = group(m, n, k, cnt, 200 + k)
    os.environ.pop("SM_FUSED_D256", None)
    ff(); torch.cuda.synchronize()
    ref = [c.clone() for c in Cs]
    tf0 = t([(ff, s2)])
    os.environ["SM_FUSED_D256"] = "1"
    ff(); torch.cuda.synchronize()
    same = all(torch.equal(a.view(torch.int16), c.view(torch.int16)) for a, c in zip(ref, Cs))
    tf1 = t([(ff, s2)])
    tiles0 = -(-m * b // 256) * -(-n // 256) * cnt if (n > 256 or True) else 0
    tiles1 = -(-m * b // 128) * -(-n // 256) * cnt
    print(f"{fn:18s} alone: product {tf0:6.1f} us | D256 {tf1:6.1f} us ({tiles1} tiles of 128 x 256)  bit-identical {same}", flush=True)
    for dn, df in direct.items():
        os.environ.pop("SM_FUSED_D256", None)
        td = t([(df, s1)])
        both0 = min(t([(df, s1), (ff, s2)]), t([(ff, s2), (df, s1)]))
        os.environ["SM_FUSED_D256"] = "1"
        both1 = min(t([(df, s1), (ff, s2)]), t([(ff, s2), (df, s1)]))
        print(f"    with {dn:18s} ({td:6.1f} us): together product {both0:6.1f} (serial {td + tf0:6.1f}, saves {100 * (1 - both0 / (td + tf0)):4.1f} %) | "
              f"together D256 {both1:6.1f} (serial {td + tf1:6.1f}, saves {100 * (1 - both1 / (td + tf1)):4.1f} %)", flush=True)
os.environ.pop("SM_FUSED_D256", None)
