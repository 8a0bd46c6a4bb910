#!/usr/bin/env python3
"""What bounds the direct kernel on its shapes ("the k = 64 single-stage shapes are C-store-bound -- ablate the epilogue and say
what bounds them"): the grouped launch bench.py makes, timed whole and with parts switched off in the tuning library (SM_DIRECT_ABLATE: 1 = no C store,
2 = no stage compute (selection + SMFMAC + LDS reads), 4 = no A / B loads; the results are wrong then -- timing only).
usage: SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/direct_ablate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
SHAPES = [(12544, 256, 64, 3), (12544, 64, 64, 1), (12544, 128, 256, 1), (3136, 128, 512, 3), (12544, 64, 256, 2), (3136, 128, 1152, 4), (12544, 64, 576, 3)]
print("# us per grouped launch (b = 32); bytes = A + B + C; 'store only' = loads and compute off, 'loads only' = store and compute off")
print("%-22s %8s | %8s %8s %8s %8s %8s %8s | %7s %7s" % ("shape x count", "roof", "full", "no store", "no comp", "no loads", "store", "loads", "A MB", "C MB"))
for (m, n, k, cnt) in SHAPES:
    b = 32
    As, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    by = cnt * (b * 2 * (m * k + m * n) + 2 * k * n)
    res = []
    for ab in (0, 1, 2, 4, 6, 3):
        os.environ["SM_DIRECT_ABLATE"] = str(ab)
        res.append(min(sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b), iters=4) for _ in range(3)) * 1e3)
    print("%-22s %8.1f | %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f | %7.1f %7.1f" % (f"{m}x{n}x{k} x{cnt}", by / 8e6, *res, cnt * b * m * k * 2 / 1e6, cnt * b * m * n * 2 / 1e6), flush=True)
    del As, Bs, Cs
os.environ.pop("SM_DIRECT_ABLATE", None)
# (a "stagger" mode -- workgroups of a CU started a third of a tile apart, SM_DIRECT_STAGGER -- was measured in session r05o and removed with the hook:
#  profiles/direct_stagger_r05o.txt)
