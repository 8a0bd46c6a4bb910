#!/usr/bin/env python3
"""Layer sweep over a shape table (the build's counterpart of the reference's examples/profiling.py:4-44,
written from scratch: it drives the C ABI in-process instead of shelling out to one binary per number).

For every row (or every unique shape with --unique) of datasets/<table>.csv it times, with HIP events on
resident, randomised inputs:  gemm (reference layout: column-major pointer-array batched, one shared B),
gemm_rm (row-major, the 2:4 kernel's layout), prune (sm_sparsify_positional on m x k, as
profiling.py:11-12 does), prune24 STRIP / TILE, check, compress, spmma.  Output: a CSV with the reference's
columns m,n,k,b,gemm,prune (ms) extended with the 2:4 stages, and a printed table with effective GF/s,
algorithmic GB/s and the fraction of the per-shape roofline max(bytes / 8 TB/s, flops / peak).
"""
import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM = 8.0e12
MFMA_F16 = 2.5e15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", default="resnet50")
    ap.add_argument("--unique", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="comma list of stages to run")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--hot", action="store_true", help="one buffer set per shape (cache-resident for small layers)")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    sm = ge.load_package()
    sm.device_check()
    dev = torch.device("cuda", 0)
    path = args.table if args.table.endswith(".csv") else os.path.join(ROOT, "datasets", args.table + ".csv")
    rows = [tuple(int(x) for x in r[:4]) for r in list(csv.reader(open(path)))[1:] if r]
    if args.unique:
        seen, uniq = set(), []
        for r in rows:
            if r not in seen:
                seen.add(r)
                uniq.append((r, rows.count(r)))
    else:
        uniq = [(r, 1) for r in rows]
    only = set(args.only.split(",")) if args.only else None

    def timeit(fn):
        # hipGraph replay: device time only (a Python/ctypes call costs ~10 us, more than the small kernels)
        return sm.graph_time_ms(fn, iters=args.reps, replays=3)

    out_rows = []
    tot = {}
    hdr = ("m", "n", "k", "b", "cnt", "stage", "ms", "effTF/s", "GB/s", "roof_us", "frac")
    print("%6s %5s %5s %3s %3s %-10s %9s %9s %8s %8s %6s" % hdr)
    h16 = args.dtype in ("f16", "bf16")
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    peak = MFMA_F16 if h16 else 157.3e12
    for (m, n, k, b), cnt in uniq:
        s = 2 if h16 else 4
        # several buffer sets cycled call by call, so that a layer whose operands fit the 256 MiB Infinity
        # Cache is not timed on cache-resident data (bench.py streams 10 GB per step; this mimics it)
        foot = b * m * k * s * 1.6 + b * m * n * s
        nbuf = 1 if args.hot else max(1, min(8, int(1.0e9 // foot) + 1))
        sets = []
        for i in range(nbuf):
            A = torch.empty(b * m * k, dtype=tdt, device=dev)
            sm.fill_uniform(A, 1234 + m + k + 17 * i, 0.0, 1.0)
            C = torch.empty(b * m * n, dtype=tdt, device=dev)
            blob = torch.empty(sm.compress24_size(m, k, s, b), dtype=torch.uint8, device=dev)
            sm.compress24(A, m, k, k, b, m * k, blob)
            Ap = torch.tensor([A.data_ptr() + s * j * m * k for j in range(b)], dtype=torch.int64, device=dev)
            Cp = torch.tensor([C.data_ptr() + s * j * m * n for j in range(b)], dtype=torch.int64, device=dev)
            sets.append(dict(A=A, A2=A.clone(), C=C, blob=blob, Ap=Ap, Cp=Cp))
        Bm = torch.empty(k * n, dtype=tdt, device=dev)
        sm.fill_uniform(Bm, 99 + n, 0.0, 1.0)
        Bp = torch.tensor([Bm.data_ptr()] * b, dtype=torch.int64, device=dev)
        mask = torch.empty(m * k, dtype=torch.int64, device=dev)
        valid = torch.zeros(1, dtype=torch.int32, device=dev)
        ctr = [0]

        def nxt():
            ctr[0] += 1
            return sets[ctr[0] % nbuf]
        flops = 2.0 * m * n * k * b
        dense_bytes = b * s * (m * k + m * n) + s * k * n
        sp_bytes = b * (m * k * s / 2 + m * k / 8 + m * n * s) + s * k * n

        def f_gemm():
            S = nxt(); sm.gemm_batched(S["Ap"], Bp, S["Cp"], m, n, k, b, args.dtype)

        def f_gemm_rm():
            S = nxt(); sm.gemm_rowmajor(S["A"], Bm, S["C"], m, n, k, batch=b)

        def f_spmma():
            S = nxt(); sm.spmma(S["blob"], Bm, S["C"], m, n, k, b, 0)

        fused_ok = k % 64 == 0 and n % 8 == 0 and h16

        def f_fused():  # the whole path in one launch; shapes the fused kernel refuses run the staged pair
            S = nxt()
            if fused_ok:
                sm.spmma_fused(S["A"], Bm, S["C"], m, n, k, batch=b)
            else:
                sm.compress24(S["A"], m, k, k, b, m * k, S["blob"])
                sm.spmma(S["blob"], Bm, S["C"], m, n, k, b, 0)

        def f_compress():
            S = nxt(); sm.compress24(S["A"], m, k, k, b, m * k, S["blob"])

        def f_prune_s():
            S = nxt(); sm.prune24(S["A2"], S["A2"], b * m, k, k, sm.PRUNE_STRIP)

        def f_prune_t():
            S = nxt(); sm.prune24(S["A2"], S["A2"], b * m, k, k, sm.PRUNE_TILE)

        def f_check():
            S = nxt(); sm.prune24_check(S["A2"], b * m, k, k, valid)

        def f_api_pc():  # the prune + check + compress of sparsifyme::spmma() in one pass (TILE, out of place from the dense A)
            S = nxt(); sm.prune24_compress24(S["A"], S["A2"], m, k, k, b, m * k, S["blob"], valid, sm.PRUNE_TILE)

        def f_pc_strip():
            S = nxt(); sm.prune24_compress24(S["A"], S["A2"], m, k, k, b, m * k, S["blob"], valid, sm.PRUNE_STRIP)

        def f_prune():
            S = nxt(); sm.sparsify(S["A2"][: m * k], mask, m, k, 0.5)
        stages = [
            ("gemm", f_gemm, flops, dense_bytes, peak),
            ("gemm_rm", f_gemm_rm, flops, dense_bytes, peak),
            ("spmma", f_spmma, flops, sp_bytes, 2 * peak if h16 else peak),
            ("fused", f_fused, flops, dense_bytes, 2 * peak),
            ("compress", f_compress, 0, b * m * k * (s + s / 2 + 1 / 8), 0),
            ("prune_s", f_prune_s, 0, 2 * b * m * k * s, 0),
            ("prune_t", f_prune_t, 0, 2 * b * m * k * s, 0),
            ("check", f_check, 0, b * m * k * s, 0),
            ("api_pc", f_api_pc, 0, b * m * k * (s + s + s / 2 + 1 / 8), 0),
            ("pc_strip", f_pc_strip, 0, b * m * k * (s + s + s / 2 + 1 / 8), 0),
            ("prune", f_prune, 0, m * k * (s + s + 8), 0),
        ]
        rec = {"m": m, "n": n, "k": k, "b": b}
        for name, fn, fl, by, pk in stages:
            if only and name not in only:
                continue
            if name in ("api_pc", "pc_strip") and not h16:
                continue
            ms = timeit(fn)
            roof = max(by / HBM, fl / pk if pk else 0.0)
            print("%6d %5d %5d %3d %3d %-10s %9.4f %9.1f %8.0f %8.1f %6.3f" %
                  (m, n, k, b, cnt, name, ms, fl / ms / 1e9, by / ms / 1e6, roof * 1e6, roof * 1e3 / ms))
            rec[name] = ms
            t = tot.setdefault(name, [0.0, 0.0, 0.0, 0.0])
            t[0] += ms * cnt; t[1] += fl * cnt; t[2] += by * cnt; t[3] += roof * cnt
        out_rows.append(rec)
        del sets
    print("---- table totals (count-weighted)")
    for name, (ms, fl, by, roof) in tot.items():
        print("%-10s %9.3f ms  %9.1f effTF/s  %8.0f GB/s  roofline %8.1f us  frac %.3f" %
              (name, ms, fl / ms / 1e9, by / ms / 1e6, roof * 1e6, roof * 1e3 / ms))
    if args.out:
        keys = ["m", "n", "k", "b"] + [k for k in out_rows[0] if k not in ("m", "n", "k", "b")]
        with open(args.out, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(out_rows)


if __name__ == "__main__":
    main()
