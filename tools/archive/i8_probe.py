#!/usr/bin/env python3
"""Device time of the int8 2:4 path on the unique ResNet-50 shapes (b = 32): sm_compress24_i8 and sm_spmma_i8
(B [n][k]); effective T-op/s = 2 m n k b / t (dense-equivalent), GB/s = algorithmic bytes / t.  tools/i8_probe.py"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
rows = [tuple(int(x) for x in r[:4]) for r in list(csv.reader(open(os.path.join(ROOT, "datasets", "resnet50.csv"))))[1:] if r]
uniq = []
for r in rows:
    if r not in [u for u, _ in uniq]:
        uniq.append((r, rows.count(r)))
tot_c = tot_m = tot_f = tot_q = tot_fq = 0.0
for (m, n, k, b), cnt in uniq:
    if k % 64:
        print(f"{m:6d} {n:5d} {k:5d}: k % 64 != 0 -- not taken by sm_spmma_i8")
        continue
    nbuf = 3
    sets = []
    for i in range(nbuf):
        A = torch.randint(-128, 128, (b * m * k,), dtype=torch.int8, device=dev)
        blob = torch.empty(sm.compress24_size(m, k, 1, b), dtype=torch.uint8, device=dev)
        sm.compress24(A, m, k, k, b, m * k, blob)
        C = torch.empty(b * m * n, dtype=torch.int32, device=dev)
        Q = torch.empty(b * m * n, dtype=torch.int8, device=dev)
        sets.append((A, blob, C, Q))
    B = torch.randint(-128, 128, (n * k,), dtype=torch.int8, device=dev)
    it = [0]
    def f_mul():
        A, blob, C, Q = sets[it[0] % nbuf]; it[0] += 1
        sm.spmma_i8(blob, B, C, m, n, k, b, 0)
    def f_q():
        A, blob, C, Q = sets[it[0] % nbuf]; it[0] += 1
        sm.spmma_i8_q(blob, B, Q, m, n, k, 2.0 ** -10, b, 0)
    def f_fq():
        A, blob, C, Q = sets[it[0] % nbuf]; it[0] += 1
        sm.spmma_fused_i8(A, B, Q, m, n, k, batch=b, scale=2.0 ** -10)
    def f_cmp():
        A, blob, C, Q = sets[it[0] % nbuf]; it[0] += 1
        sm.compress24(A, m, k, k, b, m * k, blob)
    tm = sm.graph_time_ms(f_mul, iters=10, replays=3) * 1e3
    tq = sm.graph_time_ms(f_q, iters=10, replays=3) * 1e3
    tc = sm.graph_time_ms(f_cmp, iters=10, replays=3) * 1e3
    tf = sm.graph_time_ms(f_fq, iters=10, replays=3) * 1e3
    tot_fq += tf * cnt
    fl = 2.0 * m * n * k * b
    by = b * (m * k * 9 / 16 + m * n * 4) + k * n
    byq = b * (m * k * 9 / 16 + m * n) + k * n
    tot_q += tq * cnt
    print(f"{m:6d} {n:5d} {k:5d} x{cnt}: spmma_i8 {tm:8.1f} us {fl / tm / 1e6:8.1f} T-op/s {by / tm / 1e3:6.0f} GB/s | int8 out {tq:7.1f} us {fl / tq / 1e6:8.1f} T-op/s {byq / tq / 1e3:6.0f} GB/s | compress_i8 {tc:7.1f} us {b * m * k * (1 + 9 / 16) / tc / 1e3:6.0f} GB/s | fused int8 out {tf:7.1f} us {(b * m * (k + n) + k * n) / tf / 1e3:6.0f} GB/s", flush=True)
    tot_m += tm * cnt; tot_c += tc * cnt; tot_f += fl * cnt
print(f"table (k % 64 == 0 layers): spmma_i8 (int32 out) {tot_m / 1e3:.3f} ms = {tot_f / tot_m / 1e6:.0f} effective T-op/s; "
      f"spmma_i8_q (int8 out) {tot_q / 1e3:.3f} ms = {tot_f / tot_q / 1e6:.0f}; compress_i8 {tot_c / 1e3:.3f} ms; "
      f"sm_spmma_fused_i8_q (dense A in, int8 out) {tot_fq / 1e3:.3f} ms = {tot_f / tot_fq / 1e6:.0f} effective T-op/s")
