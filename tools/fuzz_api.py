#!/usr/bin/env python3
"""Randomised shapes for the no-blob API sequence (round 6): sm_prune24_spmma_{f16,bf16} -- one kernel for n <= 128, the prune + flag pass followed by the
fused kernel on the pruned operand elsewhere, the span forms for ragged k -- must leave dA equal to sm_prune24 (TILE or STRIP; per batch matrix when a
4 x 4 tile would straddle two) and dC equal to sm_spmma(sm_compress24(pruned)) BIT FOR BIT, the flag clear, and a refused call must leave A untouched;
small operands are also held against the ORACLE's prune.  Shared and per-batch B, in place and out of place, alpha / beta.  tools/fuzz_api.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sm = ge.load_package()
orc = ge.load_oracle()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
t0 = time.time()
n_ok = n_ref = n_oracle = 0
forms = {}
while time.time() - t0 < budget:
    m = int(rng.choice([4, 6, 34, 96, 130, 132, 196, 258, 784, 1000, 3136]))
    n = int(rng.choice([8, 24, 64, 72, 128, 136, 256, 264, 512, 520, 1024, 2048, 60, 4]))
    k = int(rng.choice([64, 128, 192, 256, 448, 512, 576, 1024, 1152, 147, 72, 100, 30]))
    b = int(rng.integers(1, 4))
    shared = bool(rng.integers(0, 2))
    alg = int(rng.integers(0, 2))
    bf = bool(rng.integers(0, 2))
    inplace = bool(rng.integers(0, 2))
    alpha, beta = ((1.0, 0.0) if rng.integers(0, 2) else (0.5, -2.0))
    if b * m * max(n, k) > 3e7:
        continue
    tdt = torch.bfloat16 if bf else torch.float16
    A = torch.randn(b * m * k, dtype=torch.float32, device=dev)
    if rng.integers(0, 2):
        A = torch.round(A * 2)                                        # ties
    A[torch.rand_like(A) < 0.15] = 0
    A = A.to(tdt)
    B = torch.randn((1 if shared else b) * k * n, dtype=torch.float32, device=dev).to(tdt)
    sB = 0 if shared else k * n
    C0 = torch.randn(b * m * n, dtype=torch.float32, device=dev).to(tdt)
    Ain = A.clone()
    Aout = Ain if inplace else torch.full_like(Ain, 7.0)
    C = C0.clone()
    valid = torch.full((1,), 5, dtype=torch.int32, device=dev)
    rc = sm.prune24_spmma(Ain, Aout, B, C, m, n, k, batch=b, strideB=sB, alg=alg, d_valid=valid, alpha=alpha, beta=beta, check=False)
    torch.cuda.synchronize()
    if rc == sm.STATUS_NOT_SUPPORTED:
        if not torch.equal(Ain.view(torch.int16), A.view(torch.int16)) or not torch.equal(C.view(torch.int16), C0.view(torch.int16)):
            print("REFUSED call modified its operands", m, n, k, b, shared, alg, bf); sys.exit(1)
        n_ref += 1
        continue
    if rc != 0:
        print("unexpected status", rc, m, n, k, b); sys.exit(1)
    # the staged sequence
    P = A.clone()
    if m % 4 == 0 or alg == 1 or b == 1:
        sm.prune24(P, P, b * m, k, k, alg)
    else:
        for i in range(b):
            sm.prune24(P[i * m * k:(i + 1) * m * k], P[i * m * k:(i + 1) * m * k], m, k, k, alg)
    blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
    sm.compress24(P, m, k, k, b, m * k, blob)
    Cref = C0.clone()
    sm.spmma(blob, B, Cref, m, n, k, b, sB, alpha=alpha, beta=beta)
    torch.cuda.synchronize()
    what = (m, n, k, b, "shared" if shared else "perbatch", "tile" if alg == 0 else "strip", "bf16" if bf else "f16", "inplace" if inplace else "oop", alpha, beta)
    if int(valid.item()) != 0:
        print("flag raised", what); sys.exit(1)
    if not torch.equal(Aout.view(torch.int16), P.view(torch.int16)):
        print("MISMATCH pruned A", what); sys.exit(1)
    if not torch.equal(C.view(torch.int16), Cref.view(torch.int16)):
        print("MISMATCH C", what); sys.exit(1)
    if not inplace and not torch.equal(Ain.view(torch.int16), A.view(torch.int16)):
        print("A_in modified by the out-of-place form", what); sys.exit(1)
    if b * m * k <= 300000:   # the pruned operand against the oracle
        hA = A.view(torch.int16).cpu().numpy().view(np.uint16)
        oalg = orc.TILE if alg == 0 else orc.STRIP
        if m % 4 == 0 or alg == 1 or b == 1:
            want = orc.prune24(hA, b * m, k, k, oalg, bf16=bf)
        else:
            want = np.concatenate([orc.prune24(hA[i * m * k:(i + 1) * m * k], m, k, k, oalg, bf16=bf) for i in range(b)])
        if not np.array_equal(P.view(torch.int16).cpu().numpy().view(np.uint16), want):
            print("MISMATCH pruned A vs ORACLE", what); sys.exit(1)
        n_oracle += 1
    key = "one kernel" if (n <= 128 and n % 8 == 0 and k % 64 == 0 and m % 4 == 0) else ("span" if k % 64 else "pass + fused")
    forms[key] = forms.get(key, 0) + 1
    n_ok += 1
print("fuzz ok:", n_ok, "shapes taken", forms, "|", n_ref, "refused (operands untouched) |", n_oracle, "also against the oracle's prune")
