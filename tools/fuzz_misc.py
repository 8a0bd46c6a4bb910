#!/usr/bin/env python3
"""Randomised shapes for the newer entry points, against the oracle:
   im2col / im2col_compress24 (bit-exact), batched GEMM with transposed operands (tolerance), bf16 fused == staged
   (bit-exact).  tools/fuzz_misc.py [seconds] [seed]"""
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
n_im = n_tr = n_bf = 0
while time.time() - t0 < budget:
    # ---- im2col
    kh, kw = int(rng.choice([1, 2, 3, 5, 7])), int(rng.choice([1, 2, 3, 5, 7]))
    s, d = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    p = int(rng.integers(0, 4))
    N, C = int(rng.integers(1, 4)), int(rng.choice([1, 3, 8, 17, 64, 100]))
    H, W = int(rng.integers(1, 40)), int(rng.integers(1, 90))
    if H + 2 * p >= d * (kh - 1) + 1 and W + 2 * p >= d * (kw - 1) + 1 and N * C * H * W * kh * kw < 4e6:
        X = rng.integers(0, 1 << 16, N * C * H * W).astype(np.uint16)
        X[(X & 0x7fff) > 0x7c00] = 0x3c00
        dX = torch.from_numpy(X.view(np.float16)).cuda()
        OH, OW = sm.conv_out_size(H, kh, s, p, d), sm.conv_out_size(W, kw, s, p, d)
        L, K = OH * OW, C * kh * kw
        want = orc.im2col(X, N, C, H, W, kh, kw, s, p, d)
        dA = torch.zeros(N * L * K, dtype=torch.float16, device="cuda")
        sm.im2col(dX, N, C, H, W, kh, kw, s, p, d, dA)
        torch.cuda.synchronize()
        assert np.array_equal(dA.cpu().numpy().view(np.uint16), want), ("im2col", N, C, H, W, kh, kw, s, p, d)
        blob = torch.zeros(sm.compress24_size(L, K, 2, N), dtype=torch.uint8, device="cuda")
        sm.im2col(dX, N, C, H, W, kh, kw, s, p, d, blob, compress=True)
        torch.cuda.synchronize()
        assert np.array_equal(blob.cpu().numpy(), orc.compress24(want, L, K, K, N)), ("im2col_compress", N, C, H, W, kh, kw, s, p, d)
        n_im += 1
    # ---- transposed operands
    m, n, k = int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 200))
    ta, tb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    if (not ta or m >= k) and (not tb or k >= n):
        for sfx, dt, tol in (("f16", np.float16, 1e-2), ("f32", np.float32, 1e-3)):
            na, nb = (m * m if ta else m * k), (k * k if tb else k * n)
            A, B, C0 = rng.uniform(-1, 1, na).astype(dt), rng.uniform(-1, 1, nb).astype(dt), rng.uniform(-1, 1, m * n).astype(dt)
            dA_, dB_, dC_ = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(C0.copy()).cuda()
            ptr = lambda t: torch.tensor([t.data_ptr()], dtype=torch.int64, device="cuda")
            sm.gemm_batched(ptr(dA_), ptr(dB_), ptr(dC_), m, n, k, 1, sfx, 0.5, 1.5, ta=ta, tb=tb)
            view = (lambda x: x.view(np.uint16)) if sfx == "f16" else (lambda x: x)
            Cs = [view(C0.copy())]
            orc.gemm_batched([view(A)], [view(B)], Cs, m, n, k, 0.5, 1.5, ta=ta, tb=tb)
            ref = (Cs[0].view(np.float16) if sfx == "f16" else Cs[0]).astype(np.float64)
            a64, b64 = np.abs(A.astype(np.float64)), np.abs(B.astype(np.float64))
            opA = a64.reshape(m, m)[:, :k] if ta else a64.reshape(k, m).T
            opB = b64.reshape(k, k)[:, :n] if tb else b64.reshape(n, k).T
            scale = 0.5 * (opA @ opB).T.reshape(-1) + 1.5 * np.abs(C0.astype(np.float64))
            torch.cuda.synchronize()
            err = np.abs(dC_.cpu().numpy().astype(np.float64) - ref)
            assert (err <= tol * np.maximum(scale, 1e-30)).all(), ("transposed", sfx, m, n, k, ta, tb, err.max())
        n_tr += 1
    # ---- bf16 fused == staged
    m, n, k, b = int(rng.choice([2, 34, 130, 196, 784, 1000])), int(rng.choice([8, 64, 72, 128, 256, 264, 512, 1024])), int(rng.choice([64, 128, 192, 512, 576, 1152])), int(rng.integers(1, 3))
    if b * m * max(n, k) < 2e7:
        A = torch.randn(b * m * k, device="cuda").to(torch.bfloat16)
        A[torch.rand(b * m * k, device="cuda") < 0.2] = 0
        B = torch.randn(k * n, device="cuda").to(torch.bfloat16)
        blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device="cuda")
        sm.compress24(A, m, k, k, b, m * k, blob)
        C1 = torch.zeros(b * m * n, dtype=torch.bfloat16, device="cuda")
        C2 = torch.ones(b * m * n, dtype=torch.bfloat16, device="cuda")
        sm.spmma(blob, B, C1, m, n, k, b, 0)
        sm.spmma_fused(A, B, C2, m, n, k, batch=b)
        torch.cuda.synchronize()
        assert torch.equal(C1.view(torch.int16), C2.view(torch.int16)), ("bf16 fused", m, n, k, b)
        n_bf += 1
print(f"fuzz ok: {n_im} im2col, {n_tr} transposed, {n_bf} bf16 fused cases")
