#!/usr/bin/env python3
"""Device time of the im2col front end on the ResNet-50 convolutions (batch 32, fp16): sm_im2col_f16 (dense A),
sm_im2col_compress24_f16 (straight to the 2:4 blob) and, for comparison, sm_compress24_f16 of the dense A.
GB/s = (bytes read + bytes written, algorithmic) / time.  tools/im2col_probe.py"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
spec = importlib.util.spec_from_file_location("gen_shapes", os.path.join(ROOT, "datasets", "gen_shapes.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
dev = torch.device("cuda", 0)
N = 32
seen, cfgs, h = set(), [], 224
for cin, cout, ksz, stride, pad in gen.resnet_convs("resnet50"):
    cfg = (cin, h, ksz, stride, pad)
    h = gen.conv_out(h, ksz, stride, pad)
    if cfg not in seen:
        seen.add(cfg); cfgs.append(cfg)
tot = [0.0, 0.0, 0.0]
print("%5s %4s %2s %2s | %6s %5s | %9s %8s | %9s %8s | %9s" % ("C", "H", "k", "s", "L", "K", "im2col us", "GB/s", "fused us", "GB/s", "compress"))
for (C, H, k, s, p) in cfgs:
    OH = sm.conv_out_size(H, k, s, p, 1)
    L, K = OH * OH, C * k * k
    X = torch.empty(N * C * H * H, dtype=torch.float16, device=dev); sm.fill_uniform(X, C + H, -1.0, 1.0)
    A = torch.empty(N * L * K, dtype=torch.float16, device=dev)
    blob = torch.empty(sm.compress24_size(L, K, 2, N), dtype=torch.uint8, device=dev)
    t1 = sm.graph_time_ms(lambda: sm.im2col(X, N, C, H, H, k, k, s, p, 1, A), iters=5, replays=3) * 1e3
    t2 = sm.graph_time_ms(lambda: sm.im2col(X, N, C, H, H, k, k, s, p, 1, blob, compress=True), iters=5, replays=3) * 1e3
    t3 = sm.graph_time_ms(lambda: sm.compress24(A, L, K, K, N, L * K, blob), iters=5, replays=3) * 1e3
    bx, ba, bb = X.numel() * 2, A.numel() * 2, blob.numel()
    print("%5d %4d %2d %2d | %6d %5d | %9.1f %8.0f | %9.1f %8.0f | %9.1f" % (C, H, k, s, L, K, t1, (bx + ba) / t1 / 1e3, t2, (bx + bb) / t2 / 1e3, t3), flush=True)
    tot[0] += t1; tot[1] += t2; tot[2] += t3
    del X, A, blob
print("unique convolutions: im2col %.0f us, im2col+compress fused %.0f us, (im2col then compress: %.0f us)" % (tot[0], tot[1], tot[0] + tot[2]))
