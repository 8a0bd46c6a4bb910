#!/usr/bin/env python3
"""(round 6: the `one-kern` column is sm_prune24_spmma_* on EVERY shape -- one kernel for n <= 128, the prune + flag pass followed by the fused kernel
on the pruned operand elsewhere; no blob in either.)  Where the time of the API-faithful sequence (sparsifyme::spmma(): TILE prune in place + check + multiply, spmma.hxx:82-113)
goes, per unique shape of a table: the two-launch form (sm_prune24_compress24 + sm_spmma) and the one-kernel form
(sm_prune24_spmma, where it applies), each launch alone on one stream, hipGraph-timed on resident operands.
bytes = what the sequence has to move: A read + pruned A written (+ blob written and read back in the two-launch form) + B + C.
usage: python tools/api_path_table.py [--table resnet50] > profiles/api_path_rNN.txt"""
import argparse
import collections
import csv
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
    path = os.path.join(ROOT, "datasets", a.table + ".csv")
    rows = [tuple(int(x) for x in r[:4]) for r in list(csv.reader(open(path)))[1:] if r]
    cnt = collections.Counter(rows)
    print(f"# {os.path.basename(path)}: {len(rows)} layers, {len(cnt)} unique shapes; library {sm.version()}; us per layer, one launch at a time")
    print("%6s %5s %5s %3s %3s | %9s %8s %8s | %9s %8s | %8s" % ("m", "n", "k", "b", "cnt", "prune+cmp", "spmma", "TB/s", "one-kern", "TB/s", "min-bytes"))
    tot = collections.defaultdict(float)
    valid = torch.zeros(1, dtype=torch.int32, device=dev)
    for (m, n, k, b), c in cnt.items():
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + m + k, -1.0, 1.0)
        Ap = torch.empty_like(A)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + n, -1.0, 1.0)
        C = torch.empty(b * m * n, dtype=torch.float16, device=dev)
        blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)

        def t(fn):
            return min(sm.graph_time_ms(fn, iters=4, replays=3) for _ in range(a.reps)) * 1e3
        t_pc = t(lambda: sm.prune24_compress24(A, Ap, m, k, k, b, m * k, blob, valid, sm.PRUNE_TILE))
        t_mul = t(lambda: sm.spmma(blob, B, C, m, n, k, b, 0))
        a_el = b * m * k
        by2 = a_el * 2 * 2 + a_el * 1.125 * 2 + 2 * k * n + b * m * n * 2
        by1 = a_el * 2 * 2 + 2 * k * n + b * m * n * 2
        one = sm.prune24_spmma(A, Ap, B, C, m, n, k, batch=b, d_valid=valid, check=False) == 0
        t_one = t(lambda: sm.prune24_spmma(A, Ap, B, C, m, n, k, batch=b, d_valid=valid)) if one else float("nan")
        print("%6d %5d %5d %3d %3d | %9.1f %8.1f %8.2f | %9.1f %8.2f | %8.1f" %
              (m, n, k, b, c, t_pc, t_mul, by2 / (t_pc + t_mul) / 1e6, t_one, by1 / t_one / 1e6 if one else float("nan"), by1 / 8e6), flush=True)
        tot["two"] += (t_pc + t_mul) * c
        tot["best"] += (t_one if one else t_pc + t_mul) * c
        tot["roof"] += by1 / 8e6 * c
        tot["one_layers"] += c if one else 0
        del A, Ap, B, C, blob
    print("# serial sums over the table (us): two-launch %.0f, one kernel where it applies (%d layers) %.0f, bytes of the one-kernel form at 8 TB/s %.0f" %
          (tot["two"], tot["one_layers"], tot["best"], tot["roof"]))


if __name__ == "__main__":
    main()
