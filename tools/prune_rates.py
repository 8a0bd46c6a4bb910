#!/usr/bin/env python3
"""Device rates of the prune-step kernels on one large operand (12544 x 576 x 32 fp16 / fp32, the table's largest A): check, STRIP prune,
TILE prune, compress, one-pass prune + check + compress; algorithmic bytes / hipGraph-timed seconds.
usage: python tools/prune_rates.py > profiles/prune_rates_rNN.txt"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
m, k, b = 12544, 576, 32
for dt, es in ((torch.float16, 2), (torch.float32, 4)):
    A = torch.empty(b * m * k, dtype=dt, device=dev); sm.fill_uniform(A, 3, -1.0, 1.0)
    P = torch.empty_like(A)
    blob = torch.empty(sm.compress24_size(m, k, es, b), dtype=torch.uint8, device=dev)
    v = torch.zeros(1, dtype=torch.int32, device=dev)
    el = b * m * k
    t = lambda fn: min(sm.graph_time_ms(fn, iters=6) for _ in range(3)) * 1e-3
    sm.prune24(A, P, b * m, k, k, sm.PRUNE_STRIP)
    Q = P.clone()   # a valid 2:4 operand: the check raises no flag
    rows = [("check (valid operand)", lambda: sm.prune24_check(Q, b * m, k, k, v), el * es),
            ("check (every strip invalid)", lambda: sm.prune24_check(A, b * m, k, k, v), el * es),
            ("prune STRIP (out of place)", lambda: sm.prune24(A, P, b * m, k, k, sm.PRUNE_STRIP), 2 * el * es),
            ("prune TILE (out of place)", lambda: sm.prune24(A, P, b * m, k, k, sm.PRUNE_TILE), 2 * el * es),
            ("compress", lambda: sm.compress24(A, m, k, k, b, m * k, blob), el * (es + es / 2 + 0.125)),
            ("one pass TILE prune + check + compress", lambda: sm.prune24_compress24(A, P, m, k, k, b, m * k, blob, v, sm.PRUNE_TILE), el * (2 * es + es / 2 + 0.125))]
    for name, fn, by in rows:
        s_ = t(fn)
        print("%-5s %-42s %8.1f us  %6.2f TB/s" % ("fp16" if es == 2 else "fp32", name, s_ * 1e6, by / s_ / 1e12), flush=True)
    del A, P, Q, blob
