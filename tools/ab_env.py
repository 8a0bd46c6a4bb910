#!/usr/bin/env python3
"""A/B of one per-call environment hook of the tuning library on grouped fused launches: usage
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/ab_env.py VAR v0,v1 m,n,k,cnt [m,n,k,cnt ...]   (three alternating rounds, b = 32)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
F32 = "--f32" in sys.argv   # the fp32 split form (planes = 3), one call per instance, instead of the grouped fp16 fused launch
argv = [a for a in sys.argv if a != "--f32"]
VAR, vals = argv[1], argv[2].split(",")
for spec in argv[3:]:
    m, n, k, cnt = (int(x) for x in spec.split(","))
    b = 32
    if F32:
        A = torch.empty(b * m * k, dtype=torch.float32, device=dev); sm.fill_uniform(A, 1, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float32, device=dev); sm.fill_uniform(B, 20, -1.0, 1.0)
        C = torch.empty(b * m * n, dtype=torch.float32, device=dev)
        ws = torch.empty(max(16, sm.spmma_fused_f32_split_workspace(n, k, planes=3)), dtype=torch.uint8, device=dev)
        res = {v: 1e9 for v in vals}
        ref = None
        for r in range(3):
            for v in vals:
                os.environ[VAR] = v
                res[v] = min(res[v], sm.graph_time_ms(lambda: sm.spmma_fused_f32_split(A, B, C, m, n, k, ws, batch=b, planes=3), iters=4) * 1e3)
                torch.cuda.synchronize()
                if ref is None: ref = C.clone()
                else: assert torch.equal(ref, C), "the variants give different C"
        print(f"fp32 split {m}x{n}x{k} (per instance): " + "  ".join(f"{VAR}={v}: {res[v]:7.1f} us" for v in vals) + "  (same C bit for bit)", flush=True)
        continue
    As, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    res = {v: [] for v in vals}
    ref = None
    for r in range(3):
        for v in vals:
            os.environ[VAR] = v
            res[v].append(min(sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b), iters=4) for _ in range(2)) * 1e3)
            torch.cuda.synchronize()
            if r == 0:
                h = [c.clone() for c in Cs]
                if ref is None: ref = h
                else: assert all(torch.equal(x, y) for x, y in zip(ref, h)), "the variants give different C"
    print(f"{m}x{n}x{k} x{cnt}: " + "  ".join(f"{VAR}={v}: {min(res[v]):7.1f} us" for v in vals) + "  (same C bit for bit)", flush=True)
