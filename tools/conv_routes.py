#!/usr/bin/env python3
"""The 3 x 3 convolution layers of the ResNet-50 table (stride 1, pad 1, N = 32) by three routes, per layer, hipGraph-timed:
  conv      sm_conv_spmma_fused (implicit GEMM from NCHW activations: neither A nor a blob in HBM)
  im2col+f  sm_im2col (activations -> dense A in a workspace) + sm_spmma_fused (2:4 selection + matmul from that A)
  im2col24  sm_im2col_compress24 (activations -> 2:4 blob) + sm_spmma (staged matmul on the blob)
All three give the same C bit for bit (checked here).  usage: python tools/conv_routes.py > profiles/conv_routes_rNN.txt"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
N = 32
print("%5s %4s %5s %3s | %8s | %8s %8s %8s | %8s %8s %8s | same C" % ("Cin", "HW", "n", "cnt", "conv", "im2col", "fused", "sum", "im2col24", "spmma", "sum"))
tot = [0.0, 0.0, 0.0, 0.0]
for Cin, HW, n, cnt in [(64, 112, 64, 3), (128, 56, 128, 4), (256, 28, 256, 6), (512, 14, 512, 3)]:
    L, K = HW * HW, Cin * 9
    X = torch.empty(N * Cin * L, dtype=torch.float16, device=dev); sm.fill_uniform(X, 7 + Cin, -1.0, 1.0)
    B = torch.empty(K * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 9 + n, -1.0, 1.0)
    C1 = torch.empty(N * L * n, dtype=torch.float16, device=dev)
    C2, C3 = torch.empty_like(C1), torch.empty_like(C1)
    A = torch.empty(N * L * K, dtype=torch.float16, device=dev)
    blob = torch.empty(sm.compress24_size(L, K, 2, N), dtype=torch.uint8, device=dev)
    t = lambda fn: min(sm.graph_time_ms(fn, iters=6) for _ in range(3)) * 1e3
    t_conv = t(lambda: sm.conv_spmma_fused(X, B, C1, N, Cin, HW, HW, 3, 3, 1, 1, 1, n))
    t_i = t(lambda: sm.im2col(X, N, Cin, HW, HW, 3, 3, 1, 1, 1, A))
    t_f = t(lambda: sm.spmma_fused(A, B, C2, L, n, K, batch=N))
    t_ic = t(lambda: sm.im2col(X, N, Cin, HW, HW, 3, 3, 1, 1, 1, blob, compress=True))
    t_m = t(lambda: sm.spmma(blob, B, C3, L, n, K, N, 0))
    torch.cuda.synchronize()
    same = torch.equal(C1.view(torch.int16), C2.view(torch.int16)) and torch.equal(C1.view(torch.int16), C3.view(torch.int16))
    print("%5d %4d %5d %3d | %8.1f | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f | %s" % (Cin, HW, n, cnt, t_conv, t_i, t_f, t_i + t_f, t_ic, t_m, t_ic + t_m, same), flush=True)
    best = min(t_conv, t_i + t_f, t_ic + t_m)
    for j, v in enumerate((t_conv, t_i + t_f, t_ic + t_m, best)):
        tot[j] += v * cnt
    del X, A, blob, C1, C2, C3
print("# table-weighted (us): conv %.0f  im2col + fused %.0f  im2col24 + spmma %.0f  best route per layer %.0f" % tuple(tot))
