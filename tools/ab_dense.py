#!/usr/bin/env python3
"""A/B of the dense fp16 GEMM with and without the dense twin (the fused 2:4 kernels' pipelines run with dense MFMA; tuning library:
SM_GEMM_TWIN = 0 never / 1 the rule / 2 wherever supported), per unique shape of a table, launched the two ways bench.py launches the
dense comparator: the `cnt` instances of a shape as ONE pointer-array call (grouped) and one instance alone.  C of every mode is
compared with mode 0's (max |diff| relative to max |C|; the pipelines may differ in summation order).
usage: SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/ab_dense.py [table] [rounds]"""
import collections
import csv
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
table = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = [tuple(int(x) for x in r[:4]) for r in list(csv.reader(open(os.path.join(ROOT, "datasets", table + ".csv"))))[1:] if r]
cnt = collections.Counter(rows)
MODES = ["0", "1", "2"]
print(f"# {table}.csv, fp16; us per instance; modes SM_GEMM_TWIN = 0 / 1 / 2; library {sm.version()}")
print("%6s %5s %5s %3s %3s | %-26s | %-26s | %s" % ("m", "n", "k", "b", "cnt", "grouped  (0 / 1 / 2)", "alone  (0 / 1 / 2)", "max rel diff of C vs mode 0 (1, 2)"))
tot = collections.defaultdict(float)
for (m, n, k, b), c in cnt.items():
    As = [torch.empty(b * m * k, dtype=torch.float16, device=dev) for _ in range(c)]
    Bs = [torch.empty(k * n, dtype=torch.float16, device=dev) for _ in range(c)]
    Cs = [torch.empty(b * m * n, dtype=torch.float16, device=dev) for _ in range(c)]
    for i in range(c):
        sm.fill_uniform(As[i], 1 + i + m + k, -1.0, 1.0)
        sm.fill_uniform(Bs[i], 20 + i + n, -1.0, 1.0)
    gA = torch.tensor([x.data_ptr() for x in As], dtype=torch.int64, device=dev)
    gB = torch.tensor([x.data_ptr() for x in Bs], dtype=torch.int64, device=dev)
    gC = torch.tensor([x.data_ptr() for x in Cs], dtype=torch.int64, device=dev)
    tg, t1, ref, diffs = collections.defaultdict(list), collections.defaultdict(list), None, []
    for r in range(rounds):
        for mode in MODES:
            os.environ["SM_GEMM_TWIN"] = mode
            if r == 0:
                Cs[0].zero_()
                sm.gemm_batched(gB, gA, gC, n, m * b, k, c, "f16")
                torch.cuda.synchronize()
                got = Cs[0].float()
                if ref is None:
                    ref = got
                else:
                    diffs.append(((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item())
            tg[mode].append(sm.graph_time_ms(lambda: sm.gemm_batched(gB, gA, gC, n, m * b, k, c, "f16"), iters=4) * 1e3 / c)
            t1[mode].append(sm.graph_time_ms(lambda: sm.gemm_rowmajor(As[0], Bs[0], Cs[0], m, n, k, batch=b), iters=4) * 1e3)
    print("%6d %5d %5d %3d %3d | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f | %s" %
          (m, n, k, b, c, *[min(tg[x]) for x in MODES], *[min(t1[x]) for x in MODES], " ".join("%.2e" % d for d in diffs)), flush=True)
    for x in MODES:
        tot["g" + x] += min(tg[x]) * c
        tot["a" + x] += min(t1[x]) * c
    del As, Bs, Cs
os.environ.pop("SM_GEMM_TWIN", None)
print("# serial sums over the table (us): grouped " + " / ".join("%.0f" % tot["g" + x] for x in MODES) + "; alone " + " / ".join("%.0f" % tot["a" + x] for x in MODES))
