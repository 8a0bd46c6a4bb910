#!/usr/bin/env python3
"""Per-shape table of the fp32 2:4 forms (BASELINE config 2, datasets/resnet18.csv): the dense fp32 GEMM, the exact fused kernel
(sm_spmma_fused_f32: dense fp32 MFMA on the selected operand) and the split forms on the sparse matrix instruction
(sm_spmma_fused_f32_split, planes = 3 / 2), one launch at a time, hipGraph-timed on resident operands; bytes = A + B + C in fp32.
usage: python tools/f32_split_table.py [--table resnet18] > profiles/f32_split_rNN.txt"""
import argparse
import collections
import csv
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", default="resnet18")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    sm = ge.load_package()
    sm.device_check()
    dev = torch.device("cuda", 0)
    rows = [tuple(int(x) for x in r[:4]) for r in list(csv.reader(open(os.path.join(ROOT, "datasets", a.table + ".csv"))))[1:] if r]
    cnt = collections.Counter(rows)
    print(f"# {a.table}.csv: {len(rows)} layers, {len(cnt)} unique shapes, fp32; library {sm.version()}; us per layer, one launch at a time" +
          (" [SM_F32_SPLIT_NW=%s]" % os.environ["SM_F32_SPLIT_NW"] if "SM_F32_SPLIT_NW" in os.environ else ""))
    print("%6s %5s %5s %3s %3s | %8s %8s | %8s %6s %8s %6s | %8s %8s | %7s" % ("m", "n", "k", "b", "cnt", "dense", "exact", "split3", "TB/s", "split2", "TB/s", "dense s3", "dense s2", "roof"))
    tot = collections.defaultdict(float)
    for (m, n, k, b), c in cnt.items():
        A = torch.empty(b * m * k, dtype=torch.float32, device=dev); sm.fill_uniform(A, 1 + m + k, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float32, device=dev); sm.fill_uniform(B, 20 + n, -1.0, 1.0)
        C = torch.empty(b * m * n, dtype=torch.float32, device=dev)
        ws = torch.empty(max(16, sm.spmma_fused_f32_split_workspace(n, k, planes=3)), dtype=torch.uint8, device=dev)

        def t(fn):
            return min(sm.graph_time_ms(fn, iters=4, replays=3) for _ in range(a.reps)) * 1e3
        t_d = t(lambda: sm.gemm_rowmajor(A, B, C, m, n, k, batch=b))
        ok_e = k % 32 == 0 and n % 4 == 0
        t_e = t(lambda: sm.spmma_fused(A, B, C, m, n, k, batch=b)) if ok_e else float("nan")
        ok = sm.spmma_fused_f32_split(A, B, C, m, n, k, ws, batch=b, check=False) == 0
        t3 = t(lambda: sm.spmma_fused_f32_split(A, B, C, m, n, k, ws, batch=b, planes=3)) if ok else float("nan")
        t2 = t(lambda: sm.spmma_fused_f32_split(A, B, C, m, n, k, ws, batch=b, planes=2)) if ok else float("nan")
        d3 = t(lambda: sm.spmma_fused_f32_split(A, B, C, m, n, k, ws, batch=b, planes=3, dense=True)) if ok else float("nan")
        d2 = t(lambda: sm.spmma_fused_f32_split(A, B, C, m, n, k, ws, batch=b, planes=2, dense=True)) if ok else float("nan")
        by = 4.0 * (b * (m * k + m * n) + k * n)
        print("%6d %5d %5d %3d %3d | %8.1f %8.1f | %8.1f %6.2f %8.1f %6.2f | %8.1f %8.1f | %7.1f" % (m, n, k, b, c, t_d, t_e, t3, by / t3 / 1e6, t2, by / t2 / 1e6, d3, d2, by / 8e6), flush=True)
        for key, v in (("dense", t_d), ("dense_s3", d3 if ok else t_d), ("dense_s2", d2 if ok else t_d), ("exact", t_e if ok_e else t_d), ("split3", t3 if ok else (t_e if ok_e else t_d)), ("split2", t2 if ok else (t_e if ok_e else t_d)), ("roof", by / 8e6)):
            tot[key] += v * c
        del A, B, C, ws
    print("# serial sums over the table (us; a shape a form does not take counted at the next form's time): " +
          "  ".join(f"{k_} {tot[k_]:.0f}" for k_ in ("dense", "exact", "split3", "split2", "dense_s3", "dense_s2", "roof")))


if __name__ == "__main__":
    main()
