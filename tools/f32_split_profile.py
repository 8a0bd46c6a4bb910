#!/usr/bin/env python3
"""Runs the fp32 forms once per layer of datasets/resnet18.csv at b = 32, eagerly, for rocprofv3 passes (kernel stats; FETCH_SIZE /
WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES): the 2:4 split form with planes = 3 (sm_spmma_fused_f32_split), the exact fused form
(sm_spmma_fused_f32) and the dense fp32 GEMM.  usage: python3 tools/f32_split_profile.py [reps]"""
import csv
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
rows = [tuple(int(x) for x in r[:4]) for r in list(csv.reader(open(os.path.join(ROOT, "datasets", "resnet18.csv"))))[1:] if r]
for (m, n, k, b) in rows:
    A = torch.empty(b * m * k, dtype=torch.float32, device=dev); sm.fill_uniform(A, 1, -1.0, 1.0)
    B = torch.empty(k * n, dtype=torch.float32, device=dev); sm.fill_uniform(B, 2, -1.0, 1.0)
    C = torch.empty(b * m * n, dtype=torch.float32, device=dev)
    ws = torch.empty(max(16, sm.spmma_fused_f32_split_workspace(n, k, planes=3)), dtype=torch.uint8, device=dev)
    for _ in range(reps):
        sm.spmma_fused_f32_split(A, B, C, m, n, k, ws, batch=b, planes=3)
        if k % 32 == 0:
            sm.spmma_fused(A, B, C, m, n, k, batch=b)
        sm.gemm_rowmajor(A, B, C, m, n, k, batch=b)
    torch.cuda.synchronize()
    del A, B, C, ws
