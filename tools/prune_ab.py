#!/usr/bin/env python3
"""A/B of a per-call environment hook of the tuning library on the prune-step kernels (STRIP / TILE prune, check, compress, the one-pass prune + check + compress):
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/prune_ab.py VAR v0,v1 [m,k ...]   (b = 32, fp16; TB/s of the bytes each kernel moves)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
VAR, vals = sys.argv[1], sys.argv[2].split(",")
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[3:]] or [(12544, 576), (3136, 1152), (784, 2304), (12544, 64), (196, 4608)]
b = 32
valid = torch.zeros(1, dtype=torch.int32, device=dev)
for (m, k) in shapes:
    A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1, -1.0, 1.0)
    O = torch.empty_like(A)
    blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
    e = b * m * k
    kernels = [("prune STRIP", lambda: sm.prune24(A, O, b * m, k, k, 1), 4 * e), ("prune TILE", lambda: sm.prune24(A, O, b * m, k, k, 0), 4 * e),
               ("check", lambda: sm.prune24_check(O, b * m, k, k, valid), 2 * e), ("compress", lambda: sm.compress24(A, m, k, k, b, m * k, blob), 2 * e + e * 9 // 8),
               ("prune+check+compress TILE", lambda: sm.prune24_compress24(A, O, m, k, k, b, m * k, blob, valid, sm.PRUNE_TILE), 4 * e + e * 9 // 8)]
    for name, fn, by in kernels:
        res = {}
        for r in range(2):
            for v in vals:
                os.environ[VAR] = v
                t = sm.graph_time_ms(fn, iters=4)
                res[v] = min(res.get(v, 1e9), t)
        print(f"{m}x{k} b={b} {name:28s} " + "  ".join(f"{VAR}={v}: {res[v] * 1e3:7.1f} us {by / res[v] / 1e9:5.2f} TB/s" for v in vals), flush=True)
