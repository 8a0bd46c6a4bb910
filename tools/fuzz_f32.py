#!/usr/bin/env python3
"""Randomised shapes for the fp32 matmuls against fp64 references: sm_gemm_rowmajor_f32, sm_gemm_batched_f32 (column
major) and sm_spmma_f32 (blob from sm_compress24_f32, checked against the oracle's blob) -- shapes on and off the
LDS-DMA fast paths (K % 32 / % 64, N % 4, odd rows).  tools/fuzz_f32.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sm = ge.load_package()
orc = ge.load_oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
cnt = 0
while time.time() - t0 < budget:
    m = int(rng.choice([1, 2, 6, 33, 64, 96, 128, 130, 196, 258, 784, 1000]))
    n = int(rng.choice([1, 4, 8, 20, 63, 64, 68, 128, 132, 256, 260, 512]))
    k = int(rng.choice([4, 32, 36, 64, 96, 128, 147, 192, 256, 320, 576]))
    b = int(rng.integers(1, 4))
    alpha, beta = (1.0, 0.0) if rng.integers(0, 2) else (0.5, -1.5)
    shared = bool(rng.integers(0, 2))
    A = rng.uniform(-1, 1, b * m * k).astype(np.float32)
    A[rng.uniform(0, 1, A.size) < 0.2] = 0
    B = rng.uniform(-1, 1, (1 if shared else b) * k * n).astype(np.float32)
    C0 = rng.uniform(-1, 1, b * m * n).astype(np.float32)
    sB = 0 if shared else k * n
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    A3, B3 = A.reshape(b, m, k).astype(np.float64), B.reshape(-1, k, n).astype(np.float64)
    Bb = lambda i: B3[0] if shared else B3[i]
    # dense row-major
    dC = torch.from_numpy(C0.copy()).cuda()
    sm.gemm_rowmajor(dA, dB, dC, m, n, k, batch=b, strideB=sB, alpha=alpha, beta=beta)
    ref = np.stack([alpha * (A3[i] @ Bb(i)) for i in range(b)]).reshape(-1) + beta * C0
    scale = np.stack([abs(alpha) * (np.abs(A3[i]) @ np.abs(Bb(i))) for i in range(b)]).reshape(-1) + abs(beta) * np.abs(C0)
    torch.cuda.synchronize()
    err = np.abs(dC.cpu().numpy().astype(np.float64) - ref)
    assert (err <= 1e-5 * np.maximum(scale, 1e-30)).all(), ("gemm_rowmajor_f32", m, n, k, b, shared, err.max())
    # 2:4
    blob = torch.empty(sm.compress24_size(m, k, 4, b), dtype=torch.uint8, device="cuda")
    sm.compress24(dA, m, k, k, b, m * k, blob)
    ob = orc.compress24(A.view(np.uint32), m, k, k, b)
    torch.cuda.synchronize()
    assert np.array_equal(blob.cpu().numpy(), ob), ("compress24_f32", m, k, b)
    P = orc.prune24(A.view(np.uint32), b * m, k, k, orc.STRIP).view(np.float32).reshape(b, m, k).astype(np.float64)
    dC = torch.from_numpy(C0.copy()).cuda()
    sm.spmma(blob, dB, dC, m, n, k, b, sB, alpha=alpha, beta=beta)
    ref = np.stack([alpha * (P[i] @ Bb(i)) for i in range(b)]).reshape(-1) + beta * C0
    torch.cuda.synchronize()
    err = np.abs(dC.cpu().numpy().astype(np.float64) - ref)
    assert (err <= 1e-5 * np.maximum(scale, 1e-30)).all(), ("spmma_f32", m, n, k, b, shared, err.max())
    # column-major batched (the reference's layout), one batch entry per matrix
    Acm = [np.ascontiguousarray(A3[i].T).reshape(-1).astype(np.float32) for i in range(b)]       # m x k column-major
    Bcm = [np.ascontiguousarray(Bb(i).T).reshape(-1).astype(np.float32) for i in range(b)]       # k x n column-major
    dAs, dBs = [torch.from_numpy(x).cuda() for x in Acm], [torch.from_numpy(x).cuda() for x in Bcm]
    dCs = [torch.zeros(m * n, dtype=torch.float32, device="cuda") for _ in range(b)]
    ptr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device="cuda")
    sm.gemm_batched(ptr(dAs), ptr(dBs), ptr(dCs), m, n, k, b, "f32")
    torch.cuda.synchronize()
    for i in range(b):
        got = dCs[i].cpu().numpy().astype(np.float64).reshape(n, m).T
        sc = np.abs(A3[i]) @ np.abs(Bb(i))
        assert (np.abs(got - A3[i] @ Bb(i)) <= 1e-5 * np.maximum(sc, 1e-30)).all(), ("gemm_batched_f32", m, n, k, i)
    cnt += 1
print(f"fuzz ok: {cnt} shapes x (dense row-major, 2:4, column-major batched)")
