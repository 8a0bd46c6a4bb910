#!/usr/bin/env python3
"""Where the stream-K launch's time goes (tuning library): one shape, grouped as bench launches it, timed under SM_SK_TG / SM_SK_WG (groups of TG row panels cut into WG slot ranges; 0 0 = the library's plan) and SM_SK_ABLATE (1 = no partial stores, 2 = no fix-up waits / loads; results are wrong then: timing only).
usage: SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/sk_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
ws = sm.spmma_fused_workspace()
for (m, n, k, cnt, gs) in [(196, 512, 4608, 3, ((1, 1), (0, 0), (2, 3), (3, 5), (5, 8))), (196, 512, 2048, 2, ((1, 1), (0, 0), (1, 2), (2, 5), (3, 7))),
                          (784, 512, 1024, 1, ((1, 1), (0, 0), (4, 5), (3, 4))), (784, 256, 2304, 6, ((1, 1), (0, 0), (7, 3), (5, 2))),
                          (196, 2048, 512, 3, ((1, 1), (0, 0), (7, 3)))]:
    b = 32
    As, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    os.environ["SM_FUSED_SK"] = "0"
    t0 = sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b, workspace=ws), iters=4) * 1e3
    print(f"{m}x{n}x{k} x{cnt}: round-4 dispatch {t0:7.1f} us", flush=True)
    os.environ["SM_FUSED_SK"] = "2"
    if k == 4608:
        os.environ["SM_SK_TG"] = "3"; os.environ["SM_SK_WG"] = "5"; os.environ["SM_SK_ABLATE"] = "0"
        for lead in (0, 2, 3, 5):
            os.environ["SM_SK_LEAD"] = str(lead)
            ts = [sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b, workspace=ws), iters=4) * 1e3 for _ in range(3)]
            print(f"   tg = 3 wg = 5 lead = {lead}: {min(ts):7.1f} us (median {sorted(ts)[1]:7.1f})", flush=True)
        os.environ.pop("SM_SK_LEAD")
    for (tg, wg) in gs:   # (0, 0): the library's own plan
        for ab in (0, 3):
            os.environ["SM_SK_TG"] = str(tg); os.environ["SM_SK_WG"] = str(wg); os.environ["SM_SK_ABLATE"] = str(ab)
            try:
                t = sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b, workspace=ws), iters=4) * 1e3
            except Exception as e:
                print(f"   tg = {tg} wg = {wg}: declined ({str(e)[:80]})", flush=True)
                break
            ws[:4096].zero_()
            print(f"   tg = {tg} wg = {wg} ablate = {ab}: {t:7.1f} us", flush=True)
    del As, Bs, Cs
