#!/usr/bin/env python3
"""Randomised shapes: sm_spmma_fused_f16 must equal sm_compress24_f16 + sm_spmma_f16 bit for bit, and the dense
row-major / column-major GEMMs must agree with torch fp32 matmul within tolerance.  tools/fuzz_fused.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
t0 = time.time()
n_ok = 0
while time.time() - t0 < budget:
    m = int(rng.choice([2, 6, 34, 96, 130, 196, 258, 784, 1000, 3136, 4100]))
    n = int(rng.choice([8, 24, 64, 72, 128, 136, 256, 264, 512, 520, 1024, 2048]))
    k = int(rng.choice([64, 128, 192, 256, 448, 512, 576, 1024, 1152]))
    b = int(rng.integers(1, 4))
    shared = bool(rng.integers(0, 2))
    if b * m * max(n, k) > 3e7:
        continue
    A = torch.randn(b * m * k, dtype=torch.float16, device=dev)
    A[torch.rand_like(A, dtype=torch.float32) < 0.2] = 0          # zeros and ties
    B = torch.randn((1 if shared else b) * k * n, dtype=torch.float16, device=dev)
    sB = 0 if shared else k * n
    blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
    sm.compress24(A, m, k, k, b, m * k, blob)
    C1 = torch.zeros(b * m * n, dtype=torch.float16, device=dev)
    sm.spmma(blob, B, C1, m, n, k, b, sB)
    C2 = torch.full_like(C1, 5.0)
    sm.spmma_fused(A, B, C2, m, n, k, batch=b, strideB=sB)
    torch.cuda.synchronize()
    if not torch.equal(C1.view(torch.int16), C2.view(torch.int16)):
        print("MISMATCH fused vs staged", m, n, k, b, shared); sys.exit(1)
    # dense row-major against torch (fp32 reference)
    C3 = torch.empty_like(C1)
    sm.gemm_rowmajor(A, B, C3, m, n, k, batch=b, strideB=sB)
    ref = torch.stack([A.view(b, m, k)[i].float() @ B.view(-1, k, n)[0 if shared else i].float() for i in range(b)])
    scale = torch.stack([A.view(b, m, k)[i].float().abs() @ B.view(-1, k, n)[0 if shared else i].float().abs() for i in range(b)])
    err = ((C3.view(b, m, n).float() - ref).abs() / (scale + 1e-6)).max().item()
    if err > 1e-2:
        print("dense rowmajor off", m, n, k, b, shared, err); sys.exit(1)
    n_ok += 1
print("fuzz ok:", n_ok, "shapes")
