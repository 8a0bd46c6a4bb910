#!/usr/bin/env python3
"""One representative layer per kernel family of the timed step, a few launches each, for rocprofv3 --pmc passes
(tools/pmc_families.sh): direct<64>, direct<128>, wide, A-stationary (fused), the staged producer/consumer 2:4 matmul,
the dense GEMM on the wide shape, and the plain copy kernel as the reference point."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
b = 32


def ops(m, n, k):
    A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + m, 0.0, 1.0)
    B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 2 + n, 0.0, 1.0)
    C = torch.empty(b * m * n, dtype=torch.float16, device=dev)
    return A, B, C


for (m, n, k) in [(12544, 64, 576), (3136, 128, 1152), (784, 256, 2304), (784, 1024, 256)]:
    A, B, C = ops(m, n, k)
    for _ in range(reps):
        sm.spmma_fused(A, B, C, m, n, k, batch=b)
    torch.cuda.synchronize()
    if (m, n, k) == (784, 256, 2304):
        for _ in range(reps):
            sm.gemm_rowmajor(A, B, C, m, n, k, batch=b)
        torch.cuda.synchronize()
    del A, B, C
m, n, k = 196, 512, 4608
A, B, C = ops(m, n, k)
blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
sm.compress24(A, m, k, k, b, m * k, blob)
for _ in range(reps):
    sm.spmma(blob, B, C, m, n, k, b, 0)
torch.cuda.synchronize()
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
sm.fill_uniform(src.view(torch.float16), 9, 0.0, 1.0)
for _ in range(reps):
    sm.copy_bytes(src, dst)
torch.cuda.synchronize()
