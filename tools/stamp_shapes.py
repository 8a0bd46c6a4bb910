#!/usr/bin/env python3
"""Cycle-stamp shares of the fused kernels (diagnostic library: make -C sparsify.me_amd stamp): one grouped launch per spec, the
stamped launchers print the per-role cycle sums on stderr.
usage: SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_stamp.so python tools/stamp_shapes.py "m,n,k,count[:ENV=V[;ENV=V]]" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
touched = set()
for spec in sys.argv[1:]:
    shape, _, envs = spec.partition(":")
    m, n, k, cnt = map(int, shape.split(","))
    for key in touched:
        os.environ.pop(key, None)
    for kv in filter(None, envs.split(";")):
        key, _, val = kv.partition("=")
        os.environ[key] = val
        touched.add(key)
    b = 32
    As, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    sys.stderr.write(f"--- {m}x{n}x{k} x{cnt} {envs}\n"); sys.stderr.flush()
    for _ in range(2):
        sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b)
    torch.cuda.synchronize()
    del As, Bs, Cs
