#!/usr/bin/env python3
"""Randomised shapes for the int8 2:4 path against the oracle, bit-exact: compress, staged and fused matmul (int32 and
requantised int8 output), shared / per-batch B, accumulate.  tools/fuzz_i8.py [seconds] [seed]"""
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
    m = int(rng.choice([2, 6, 34, 96, 128, 130, 196, 258, 784, 1000]))
    n = int(rng.choice([1, 3, 8, 16, 24, 63, 64, 72, 128, 136, 256, 264, 520]))
    k = int(rng.choice([64, 128, 192, 256, 320, 576, 1024]))
    b = int(rng.integers(1, 4))
    shared = bool(rng.integers(0, 2))
    acc = bool(rng.integers(0, 2))
    A = rng.integers(-128, 128, b * m * k).astype(np.int8)
    A[rng.uniform(0, 1, A.size) < 0.25] = 0
    B = rng.integers(-128, 128, (1 if shared else b) * n * k).astype(np.int8)
    sB = 0 if shared else n * k
    C0 = rng.integers(-500, 500, b * m * n).astype(np.int32)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    blob = torch.empty(sm.compress24_size(m, k, 1, b), dtype=torch.uint8, device="cuda")
    sm.compress24(dA, m, k, k, b, m * k, blob)
    ob = orc.compress24(A.view(np.uint8), m, k, k, b)
    torch.cuda.synchronize()
    assert np.array_equal(blob.cpu().numpy(), ob), ("compress_i8", m, k, b)
    Cref = C0.copy()
    orc.spmma_i8(ob, B, Cref, m, n, k, b, sB, accumulate=acc)
    dC, dF = torch.from_numpy(C0.copy()).cuda(), torch.from_numpy(C0.copy()).cuda()
    sm.spmma_i8(blob, dB, dC, m, n, k, b, sB, accumulate=acc)
    sm.spmma_fused_i8(dA, dB, dF, m, n, k, batch=b, strideB=sB, accumulate=acc)
    torch.cuda.synchronize()
    assert np.array_equal(dC.cpu().numpy(), Cref), ("spmma_i8", m, n, k, b, shared, acc)
    assert np.array_equal(dF.cpu().numpy(), Cref), ("spmma_fused_i8", m, n, k, b, shared, acc)
    Cacc = np.zeros(b * m * n, dtype=np.int32)
    orc.spmma_i8(ob, B, Cacc, m, n, k, b, sB)
    scale = float(rng.choice([2.0 ** -7, 0.003, 1.0]))
    dQ, dQ2 = torch.empty(b * m * n, dtype=torch.int8, device="cuda"), torch.empty(b * m * n, dtype=torch.int8, device="cuda")
    sm.spmma_i8_q(blob, dB, dQ, m, n, k, scale, b, sB)
    sm.spmma_fused_i8(dA, dB, dQ2, m, n, k, batch=b, strideB=sB, scale=scale)
    torch.cuda.synchronize()
    want = orc.requant_i8(Cacc, scale)
    assert np.array_equal(dQ.cpu().numpy(), want) and np.array_equal(dQ2.cpu().numpy(), want), ("requant", m, n, k, b, scale)
    cnt += 1
print(f"fuzz ok: {cnt} int8 shapes")
