#!/usr/bin/env python3
"""Device time of sm_spmma_fused_f16 on one shape, alone and as a grouped launch of `g` instances (hipGraph-timed, cycling
buffer sets): tools/time_fused.py m n k b [g]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
m, n, k, b = map(int, sys.argv[1:5])
g = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dev = torch.device("cuda", 0)
As, Bs, Cs = [], [], []
for i in range(g):
    A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, 0.0, 1.0)
    B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, 0.0, 1.0)
    As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
state = {"i": 0}
def one():
    i = state["i"] % g; state["i"] += 1
    sm.spmma_fused(As[i], Bs[i], Cs[i], m, n, k, batch=b)
t1 = sm.graph_time_ms(one, iters=4 * g) * 1e3
tg = sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b), iters=4) * 1e3
by = b * 2 * (m * k + m * n) + 2 * k * n
print(f"{m}x{n}x{k} b={b}: single {t1:7.1f} us ({by / t1 / 1e6:5.2f} TB/s)   grouped x{g} {tg:7.1f} us = {tg / g:6.1f} per instance ({by * g / tg / 1e6:5.2f} TB/s)   roof {by / 8e6:5.1f} us")
