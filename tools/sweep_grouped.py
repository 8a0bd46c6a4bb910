#!/usr/bin/env python3
"""Per-shape table of a round, launched the way bench.py's step launches (the build's counterpart of the reference's
examples/compare.csv, examples/profiling.py:4-44): for every unique (m, n, k, b) of a shape table, with its instance
count `cnt`, the time PER INSTANCE of
  dense   sm_gemm_batched_f16 on the cnt instances as ONE pointer-array call (row-major product of the stacked operand:
          the dense comparator grouped like the step), and each instance alone (dense1)
  staged  sm_compress24_f16 + sm_spmma_f16 per instance (compress / spmma columns)
  fused   sm_spmma_fused_f16_grouped on the cnt instances (one grid per <= 8; with a stream-K workspace, as bench.py's step
          launches since round 5), and each instance alone without one (fused1)
  prune   the reference's `prune` column (examples/profiling.py:10-13, examples/compare.csv): sparsifyme::sparsify<2,2> on ONE m x k
          fp32 matrix + its 8-byte mask (what `./bin/sparsify m k` times; a1 of SURVEY.md 8), cycling buffers; pruneB: the same
          positional operator on the layer's whole fp16 operand (b * m x k + mask); each with its fraction of the 8 TB/s peak (fp32: 4 + 4 + 8 = 16 B / element, fp16: 2 + 2 + 8 = 12)
hipGraph-timed on resident random operands; roofline = max(algorithmic bytes / 8 TB/s, dense-equivalent flops / 5 PF/s).
usage: python tools/sweep_grouped.py [--table resnet50] [--reps 3] > profiles/sweep_rNN_f16_<table>.txt"""
import argparse
import collections
import csv
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fused_variant(n, k, m, b, c):
    sys.path.insert(0, ROOT)
    import bench
    return bench.fused_variant(n, k, m, b, min(8, c))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", default="resnet50")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    sm = ge.load_package()
    sm.device_check()
    dev = torch.device("cuda", 0)
    path = a.table if a.table.endswith(".csv") else os.path.join(ROOT, "datasets", a.table + ".csv")
    rows = [tuple(int(x) for x in r[:4]) for r in list(csv.reader(open(path)))[1:] if r]
    cnt = collections.Counter(rows)
    print(f"# {os.path.basename(path)}: {len(rows)} layers, {len(cnt)} unique shapes; library {sm.version()}; times in us PER INSTANCE")
    print("%6s %5s %5s %3s %3s %-7s | %8s %8s | %8s %8s | %8s %8s | %7s %6s %6s %7s | %7s %5s %7s %5s" %
          ("m", "n", "k", "b", "cnt", "kernel", "dense", "dense1", "compress", "spmma", "fused", "fused1", "roof", "frac", "TB/s", "effTF/s", "prune", "frac", "pruneB", "frac"))
    ws = sm.spmma_fused_workspace()
    tot = collections.defaultdict(float)
    for (m, n, k, b), c in cnt.items():
        As, Bs, Cs = [], [], []
        for i in range(c):
            A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i + m + k, 0.0, 1.0)
            B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i + n, 0.0, 1.0)
            As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
        blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
        gA = torch.tensor([x.data_ptr() for x in As], dtype=torch.int64, device=dev)
        gB = torch.tensor([x.data_ptr() for x in Bs], dtype=torch.int64, device=dev)
        gC = torch.tensor([x.data_ptr() for x in Cs], dtype=torch.int64, device=dev)
        st = {"i": 0}

        def nxt():
            st["i"] += 1
            return st["i"] % c
        fused_ok = (n % 8 == 0 and (k % 64 == 0 or (n <= 128 and (b * m * k * 2) % 16 == 0))) or (n < 8 and k <= 64 and (b * m * k * 2) % 16 == 0)

        def t(fn, per):
            return min(sm.graph_time_ms(fn, iters=max(2, 8 // per), replays=3) for _ in range(a.reps)) * 1e3 / per
        t_dense = t(lambda: sm.gemm_batched(gB, gA, gC, n, m * b, k, c, "f16"), c)
        t_dense1 = t(lambda: (lambda i: sm.gemm_rowmajor(As[i], Bs[i], Cs[i], m, n, k, batch=b))(nxt()), 1)
        t_cmp = t(lambda: sm.compress24(As[nxt()], m, k, k, b, m * k, blob), 1)
        sm.compress24(As[0], m, k, k, b, m * k, blob)
        t_mul = t(lambda: (lambda i: sm.spmma(blob, Bs[i], Cs[i], m, n, k, b, 0))(nxt()), 1)
        if fused_ok:   # (the library has the last word: e.g. a ragged k whose 128-row span + B do not fit the LDS stays on the staged pair)
            try:
                sm.spmma_fused(As[0], Bs[0], Cs[0], m, n, k, batch=b)
            except sm.SparsifymeError:
                fused_ok = False
        if fused_ok:
            t_fused = t(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b, workspace=ws), c)
            t_fused1 = t(lambda: (lambda i: sm.spmma_fused(As[i], Bs[i], Cs[i], m, n, k, batch=b))(nxt()), 1)
        else:
            t_fused = t_fused1 = t_cmp + t_mul
        # a1: the positional operator, as the reference's harness times it (one m x k fp32 matrix) and on the layer's whole operand
        nb32 = max(2, min(64, (256 << 20) // (m * k * 12)))   # enough buffers that small matrices are not served by the Infinity Cache
        W32 = [torch.empty(m * k, dtype=torch.float32, device=dev) for _ in range(nb32)]
        M32 = [torch.empty(m * k, dtype=torch.int64, device=dev) for _ in range(nb32)]
        for w_ in W32:
            sm.fill_uniform(w_, 7, 0.0, 1.0)
        sp = {"i": 0}

        def prune_one():
            sp["i"] = (sp["i"] + 1) % nb32
            sm.sparsify(W32[sp["i"]], M32[sp["i"]], m, k)
        t_prune = t(prune_one, 1)
        del W32, M32
        MB = torch.empty(b * m * k, dtype=torch.int64, device=dev)
        WB = As[0].clone()
        t_pruneB = t(lambda: sm.sparsify(WB, MB, b * m, k), 1)
        del MB, WB
        by = b * 2 * (m * k + m * n) + 2 * k * n
        fl = 2.0 * m * n * k * b
        roof = max(by / 8e12, fl / 5e15) * 1e6
        print("%6d %5d %5d %3d %3d %-7s | %8.1f %8.1f | %8.1f %8.1f | %8.1f %8.1f | %7.1f %6.3f %6.2f %7.0f | %7.1f %5.3f %7.1f %5.3f" %
              (m, n, k, b, c, fused_variant(n, k, m, b, c) if fused_ok else "staged", t_dense, t_dense1, t_cmp, t_mul, t_fused, t_fused1, roof, roof / t_fused,
               by / t_fused / 1e6, fl / t_fused / 1e6, t_prune, m * k * 16.0 / t_prune / 8e6, t_pruneB, b * m * k * 12.0 / t_pruneB / 8e6), flush=True)
        for key, v in (("dense", t_dense), ("dense1", t_dense1), ("compress", t_cmp), ("spmma", t_mul), ("fused", t_fused), ("fused1", t_fused1), ("roof", roof),
                       ("prune", t_prune), ("pruneB", t_pruneB)):
            tot[key] += v * c
        tot["bytes"] += by * c
        tot["flops"] += fl * c
        del As, Bs, Cs, blob
    print("# table totals, serial sum over the %d layers (us): " % len(rows) + "  ".join(f"{k_} {tot[k_]:.0f}" for k_ in ("dense", "dense1", "compress", "spmma", "fused", "fused1", "roof", "prune", "pruneB")))
    print("# fused (grouped): %.2f TB/s of algorithmic bytes = %.3f of the 8 TB/s peak, %.0f effective TF/s; dense (grouped) / fused = %.3f; dense / (2:4 matmul alone) = %.3f" %
          (tot["bytes"] / tot["fused"] / 1e6, tot["bytes"] / tot["fused"] / 8e6, tot["flops"] / tot["fused"] / 1e6, tot["dense"] / tot["fused"], tot["dense"] / tot["spmma"]))


if __name__ == "__main__":
    main()
