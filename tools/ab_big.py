#!/usr/bin/env python3
"""A/B of the fused kernel variants in ONE process (tuning library: the SM_* hooks are read per call): for every shape of
the list the grouped launch the bench step makes (count instances, b = 32) is timed under each environment setting,
interleaved over `rounds`, and the first setting's C is compared bit for bit with every other's.
usage: SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/ab_big.py [shape-set] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)

SETS = {
    "wide": [(784, 256, 2304, 6), (784, 256, 1024, 5), (3136, 256, 512, 1), (196, 512, 4608, 3), (196, 512, 2048, 2), (784, 512, 1024, 1)],
    "n128": [(3136, 128, 1152, 4), (3136, 128, 512, 3), (12544, 128, 256, 1)],
    "k64": [(12544, 256, 64, 3)],
    "astat": [(3136, 512, 128, 4), (784, 1024, 256, 6), (196, 2048, 512, 3)],
    "big2": [(784, 256, 1024, 5), (3136, 256, 512, 1), (196, 512, 4608, 3), (784, 512, 1024, 1), (196, 2048, 512, 3), (784, 256, 2304, 6), (196, 512, 2048, 2), (784, 1024, 256, 6)],
    "ilv": [(784, 256, 2304, 6), (784, 256, 1024, 5), (3136, 256, 512, 1), (196, 512, 4608, 3), (196, 512, 2048, 2), (784, 512, 1024, 1), (196, 2048, 512, 3)],
}
VARIANTS = {
    "wide": [("base", {}), ("big", {"SM_FUSED_BIG": "1"})],
    "n128": [("base", {}), ("big nsb2", {"SM_FUSED_BIG": "2", "SM_FUSED_BIG_NSB": "2"}), ("big nsb3", {"SM_FUSED_BIG": "2", "SM_FUSED_BIG_NSB": "3"})],
    "k64": [("base", {}), ("big", {"SM_FUSED_BIG": "4"})],
    "astat": [("base", {}), ("big", {"SM_FUSED_BIG": "1", "SM_FUSED_ASTAT": "0"})],
    # round 5: the software-pipelined big form (selection of stage kt + 1 interleaved into the B sweep of stage kt) against the round-4 big form,
    # under the dispatch rule and forced on every n > 128 shape
    # (big2, the software-pipelined form, measured 3-6 % slower and was removed: profiles/ab_big2_r05k.txt)
    # round 5: the split-role big form at every tile height (SM_FUSED_BIG3 = BM) against the dispatch rule
    # (round 5: big2 = the software-pipelined big form, big3 = the split-role form at five tile heights: measured, not adopted, removed from the
    #  source -- profiles/ab_big2_r05k.txt, ab_big3_r05l.txt; the shape set "big2" stays for re-use)
    "big2": [("rule", {})],
}
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["wide", "n128", "k64"]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
KEYS = sorted({k for vs in VARIANTS.values() for _, e in vs for k in e})


def setenv(e):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(e)


for setname in which:
    for (m, n, k, cnt) in SETS[setname]:
        b = 32
        As, Bs, Cs = [], [], []
        for i in range(cnt):
            A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
            B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
            As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
        by = cnt * (b * 2 * (m * k + m * n) + 2 * k * n)
        ref, times = None, {}
        for r in range(rounds):
            for name, e in VARIANTS[setname]:
                setenv(e)
                if r == 0:
                    for C in Cs:
                        C.fill_(float("nan"))
                    sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b)
                    torch.cuda.synchronize()
                    got = [C.view(torch.int16).clone() for C in Cs]
                    if ref is None:
                        ref = got
                    else:
                        same = all(torch.equal(x, y) for x, y in zip(ref, got))
                        print(f"   {m}x{n}x{k} x{cnt} [{name}] C bit-identical to [{VARIANTS[setname][0][0]}]: {same}", flush=True)
                        if not same:
                            d = [(x != y).sum().item() for x, y in zip(ref, got)]
                            print("      differing elements per instance:", d, flush=True)
                t = sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b), iters=4) * 1e3
                times.setdefault(name, []).append(t)
        row = "  ".join(f"{name}: {min(ts):7.1f} us ({by / min(ts) / 1e6:5.2f} TB/s, med {sorted(ts)[len(ts) // 2]:7.1f})" for name, ts in times.items())
        print(f"{m}x{n}x{k} b={b} x{cnt}  roof {by / 8e6:6.1f} us | {row}", flush=True)
        del As, Bs, Cs, ref
setenv({})
