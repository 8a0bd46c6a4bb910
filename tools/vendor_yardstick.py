#!/usr/bin/env python3
"""Yardstick only (never linked or called by the library, the tests or bench.py): times torch.matmul -- the vendor
BLAS behind PyTorch-ROCm -- next to sm_gemm_rowmajor_f16 on the unique ResNet-50 shapes at b = 32, to show that the
dense comparator of the 2:4 speed-up is not a straw man.  Output kept in profiles/vendor_yardstick_r01.txt."""
import csv, os, sys, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
rows = [tuple(int(x) for x in r[:4]) for r in list(csv.reader(open("datasets/resnet50.csv")))[1:] if r]
uniq = {}
for r in rows: uniq[r[:3]] = uniq.get(r[:3], 0) + 1
tot_t = tot_m = 0.0
for (m, n, k), cnt in uniq.items():
    b = 32
    A = torch.randn(b * m, k, dtype=torch.float16, device=dev)
    B = torch.randn(k, n, dtype=torch.float16, device=dev)
    C = torch.empty(b * m, n, dtype=torch.float16, device=dev)
    f = lambda: torch.matmul(A, B, out=C)
    t = sm.graph_time_ms(f, iters=10)
    Af, Bf, Cf = A.reshape(-1), B.reshape(-1), C.reshape(-1)
    g = lambda: sm.gemm_rowmajor(Af, Bf, Cf, m, n, k, batch=b)
    t2 = sm.graph_time_ms(g, iters=10)
    tot_t += t * cnt; tot_m += t2 * cnt
    print(f"{m:6d} {n:5d} {k:5d} x{cnt}  torch {t:.4f} ms  mine {t2:.4f} ms  ratio {t2/t:.2f}", flush=True)
print("total torch %.3f ms, mine %.3f ms" % (tot_t, tot_m))
