#!/usr/bin/env python3
"""A/B of the stream-K form (round 5) in ONE process: for every shape the grouped launch the bench step makes (count instances,
b = 32) is timed without a workspace (the round-4 dispatch) and with one (the library may take spmma_f16_fused_sk_kernel),
interleaved over `rounds`; the two results are compared (max |difference| in fp16 ulps of the larger value; equal elements) and the
with-workspace result is checked to be bitwise reproducible.
usage: python tools/ab_streamk.py [rounds]      (tuning library + SM_FUSED_SK=2: the stream-K kernel wherever it takes the shape)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
SHAPES = [(196, 512, 4608, 3), (196, 512, 2048, 2), (784, 512, 1024, 1), (196, 2048, 512, 3), (784, 256, 2304, 6), (784, 256, 1024, 5), (3136, 256, 512, 1),
          (784, 1024, 256, 6), (3136, 512, 128, 4)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ws = sm.spmma_fused_workspace()
for (m, n, k, cnt) in SHAPES:
    b = 32
    As, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    by = cnt * (b * 2 * (m * k + m * n) + 2 * k * n)
    res, times = {}, {}
    for r in range(rounds):
        for name, w in (("base", None), ("ws", ws)):
            if r == 0:
                for C in Cs:
                    C.fill_(float("nan"))
                ws[4096:].fill_(0xff)
                sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b, workspace=w)
                torch.cuda.synchronize()
                res[name] = [C.clone() for C in Cs]
                if w is not None:
                    used = bool((ws[4096:4096 + (64 << 20)] != 0xff).any().item())
                    flags0 = bool((ws[:4096] == 0).all().item())
                    sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b, workspace=w)
                    torch.cuda.synchronize()
                    rep = all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(res[name], Cs))
                    x, y = torch.cat(res["base"]).float(), torch.cat(res["ws"]).float()
                    neq = int((x != y).sum().item())
                    rel = float(((x - y).abs() / torch.maximum(x.abs(), y.abs()).clamp_min(1e-3)).max().item())
                    print(f"   {m}x{n}x{k} x{cnt}: stream-K ran: {used}; flags zero after: {flags0}; reproducible: {rep}; elements differing from base: {neq} of {x.numel()}"
                          f" (max relative difference {rel:.2e})", flush=True)
            t = sm.graph_time_ms(lambda: sm.spmma_fused_grouped(As, Bs, Cs, m, n, k, batch=b, workspace=w), iters=4) * 1e3
            times.setdefault(name, []).append(t)
    # the dense twin: the instances as one pointer-array call of the dense GEMM (the grouped dense comparator of bench.py), with / without
    gA = torch.tensor([x.data_ptr() for x in As], dtype=torch.int64, device=dev)
    gB = torch.tensor([x.data_ptr() for x in Bs], dtype=torch.int64, device=dev)
    gC = torch.tensor([x.data_ptr() for x in Cs], dtype=torch.int64, device=dev)
    for r in range(rounds):
        for name, w in (("dense", None), ("dense ws", ws)):
            t = sm.graph_time_ms(lambda: sm.gemm_batched(gB, gA, gC, n, m * b, k, cnt, "f16", workspace=w), iters=4) * 1e3
            times.setdefault(name, []).append(t)
    row = "  ".join(f"{name}: {min(ts):7.1f} us ({by / min(ts) / 1e6:5.2f} TB/s = {by / min(ts) / 8e6:5.3f}, med {sorted(ts)[len(ts) // 2]:7.1f})" for name, ts in times.items())
    print(f"{m}x{n}x{k} b={b} x{cnt}  roof {by / 8e6:6.1f} us | {row}", flush=True)
    del As, Bs, Cs, res
