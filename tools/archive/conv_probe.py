#!/usr/bin/env python3
"""conv_probe.py -- the 3x3 convolution layers of the ResNet-50 table (b = 32; stride 1, padding 1: m = H*W) three ways:
  implicit : sm_conv_spmma_fused_f16 straight from the NCHW activations (no dense A, no blob)
  fused    : sm_spmma_fused_f16 from a materialised dense A (the bench step's path; A assumed to exist)
  staged   : sm_im2col_compress24_f16 + sm_spmma_f16 (activations -> blob -> matmul)
  im2col+  : sm_im2col_f16 + sm_spmma_fused_f16 (activations -> dense A -> fused matmul)
Times by hipGraph replay; bytes: implicit = X + B + C, fused = A + B + C."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as ge
    sm = ge.load_package()
    sm.device_check()
    dev = torch.device("cuda", 0)
    N = 32
    layers = [(64, 112, 64, 3), (128, 56, 128, 4), (256, 28, 256, 6), (512, 14, 512, 3)]  # Cin, H = W, n_out, count in the table
    print("%5s %4s %5s %3s | %10s %10s %10s %10s | %8s %8s %7s" % ("Cin", "HW", "n", "cnt", "implicit", "fused(A)", "staged", "im2col+f", "impl TF/s", "impl GB/s", "frac"))
    tot = [0.0] * 4
    for Cin, HW, n, cnt in layers:
        L, K = HW * HW, Cin * 9
        X = torch.empty(N * Cin * L, dtype=torch.float16, device=dev)
        sm.fill_uniform(X, 7 + Cin, -1.0, 1.0)
        B = torch.empty(K * n, dtype=torch.float16, device=dev)
        sm.fill_uniform(B, 9 + n, -1.0, 1.0)
        C = torch.empty(N * L * n, dtype=torch.float16, device=dev)
        A = torch.empty(N * L * K, dtype=torch.float16, device=dev)
        blob = torch.empty(sm.compress24_size(L, K, 2, N), dtype=torch.uint8, device=dev)
        sm.im2col(X, N, Cin, HW, HW, 3, 3, 1, 1, 1, A)
        t_imp = sm.graph_time_ms(lambda: sm.conv_spmma_fused(X, B, C, N, Cin, HW, HW, 3, 3, 1, 1, 1, n))
        if os.environ.get("CONV_PROBE_ONLY") == "implicit":
            t_fus = t_stg = t_i2f = float("nan")
        else:
            t_fus = sm.graph_time_ms(lambda: sm.spmma_fused(A, B, C, L, n, K, batch=N))
            t_stg = sm.graph_time_ms(lambda: (sm.im2col(X, N, Cin, HW, HW, 3, 3, 1, 1, 1, blob, compress=True), sm.spmma(blob, B, C, L, n, K, N, 0)))
            t_i2f = sm.graph_time_ms(lambda: (sm.im2col(X, N, Cin, HW, HW, 3, 3, 1, 1, 1, A), sm.spmma_fused(A, B, C, L, n, K, batch=N)))
        fl = 2.0 * N * L * n * K
        by = 2.0 * (N * Cin * L + K * n + N * L * n)
        roof = max(by / 8e12, fl / 5e15)
        print("%5d %4d %5d %3d | %9.1fus %9.1fus %9.1fus %9.1fus | %8.1f %8.0f %7.3f" %
              (Cin, HW, n, cnt, t_imp * 1e3, t_fus * 1e3, t_stg * 1e3, t_i2f * 1e3, fl / t_imp / 1e9, by / t_imp / 1e6, roof * 1e3 / t_imp))
        for i, t in enumerate((t_imp, t_fus, t_stg, t_i2f)):
            tot[i] += t * cnt
        del X, A, blob, C
    print("table-weighted totals (ms): implicit %.3f  fused(A) %.3f  staged %.3f  im2col+fused %.3f" % tuple(tot))


if __name__ == "__main__":
    main()
