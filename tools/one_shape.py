#!/usr/bin/env python3
"""Runs one stage on one shape a few times (for rocprofv3 --pmc passes): tools/one_shape.py m n k b stage reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
m, n, k, b = map(int, sys.argv[1:5])
stage = sys.argv[5] if len(sys.argv) > 5 else "spmma"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda", 0)
A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1, 0.0, 1.0)
B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 2, 0.0, 1.0)
C = torch.empty(b * m * n, dtype=torch.float16, device=dev)
blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
sm.compress24(A, m, k, k, b, m * k, blob)
for _ in range(reps):
    if stage == "spmma":
        sm.spmma(blob, B, C, m, n, k, b, 0)
    elif stage == "gemm_rm":
        sm.gemm_rowmajor(A, B, C, m, n, k, batch=b)
    elif stage == "fused":
        sm.spmma_fused(A, B, C, m, n, k, batch=b)
    elif stage == "compress":
        sm.compress24(A, m, k, k, b, m * k, blob)
torch.cuda.synchronize()
