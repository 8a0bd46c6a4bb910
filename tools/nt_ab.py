#!/usr/bin/env python3
"""A/B of the non-temporal hint on the direct kernel's A loads (tuning library, SM_DIRECT_NT), per shape, grouped as bench.py launches, three alternating rounds.
usage: SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/nt_ab.py [SM_DIRECT_NT | SM_BIG_NT]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
VAR = sys.argv[1] if len(sys.argv) > 1 else "SM_DIRECT_NT"
SETS = {"SM_DIRECT_NT": [(12544, 256, 64, 3), (12544, 64, 64, 1), (12544, 128, 256, 1), (3136, 128, 512, 3), (12544, 64, 256, 2), (3136, 128, 1152, 4), (12544, 64, 576, 3)],
        "SM_BIG_NT": [(784, 256, 1024, 5), (3136, 256, 512, 1), (196, 512, 2048, 2)]}   # (SM_BIG_NT: measured indifferent in session r05r, hook removed)
for (m, n, k, cnt) in SETS[VAR]:
    b = 32
    As, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    for r in range(3):
        for nt in ("1", "0"):
            os.environ[VAR] = nt
            t = min(sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b), iters=4) for _ in range(2)) * 1e3
            print(f"{m}x{n}x{k} x{cnt} nt={nt}: {t:7.1f} us", flush=True)
