#!/usr/bin/env python3
"""One representative layer per kernel family of the timed step, a few launches each, for rocprofv3 --pmc passes
(tools/pmc_families.sh): direct<64>, direct<128>, wide, big (256-row tiles), A-stationary, span (fused; grouped launches since round 4),
the staged producer/consumer 2:4 matmul,
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


# round 4: every family as the timed step launches it -- the grouped launch of the shape's instance count
for (m, n, k, cnt) in [(12544, 64, 576, 3), (3136, 128, 1152, 4), (784, 256, 2304, 6), (784, 256, 1024, 5), (196, 512, 4608, 3), (784, 1024, 256, 6), (12544, 64, 147, 1)]:
    ops_ = [ops(m, n, k) for _ in range(cnt)]
    for _ in range(reps):
        sm.spmma_fused_grouped([o[0] for o in ops_], [o[1] for o in ops_], [o[2] for o in ops_], m, n, k, batch=b)
    torch.cuda.synchronize()
    if (m, n, k) == (784, 256, 2304):
        A, B, C = ops_[0]
        for _ in range(reps):
            sm.gemm_rowmajor(A, B, C, m, n, k, batch=b)
        torch.cuda.synchronize()
    del ops_
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
