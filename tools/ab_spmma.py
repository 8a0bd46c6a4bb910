#!/usr/bin/env python3
"""A/B of the staged 2:4 matmul (sm_spmma_f16 on prepared blobs) under tuning switches, per few-tile shape of the ResNet-50 table,
launched grouped (sm_spmma_f16_grouped over the shape's instances) as bench.py's grouped stage does; C compared bit for bit.
usage: SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so python tools/ab_spmma.py [rounds]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
SHAPES = [(784, 256, 2304, 6), (784, 256, 1024, 5), (3136, 256, 512, 1), (3136, 512, 128, 4), (784, 1024, 256, 6), (784, 512, 1024, 1), (196, 512, 4608, 3),
          (196, 2048, 512, 3), (196, 512, 2048, 2), (12544, 256, 64, 3)]
VARIANTS = [("base", {}), ("big ring 3", {"SM_SPMMA_BIG": "3"}), ("big ring 2", {"SM_SPMMA_BIG": "2"})]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
KEYS = sorted({k for _, e in VARIANTS for k in e})
tot = {name: 0.0 for name, _ in VARIANTS}
for (m, n, k, cnt) in SHAPES:
    b = 32
    blobs, Bs, Cs = [], [], []
    for i in range(cnt):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
        blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
        sm.compress24(A, m, k, k, b, m * k, blob)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
        blobs.append(blob); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
        del A
    by = cnt * (b * (m * k * 9 // 8 + 2 * m * n) + 2 * k * n)
    ref, times = None, {}
    for r in range(rounds):
        for name, e in VARIANTS:
            for key in KEYS:
                os.environ.pop(key, None)
            os.environ.update(e)
            if r == 0:
                for C in Cs:
                    C.fill_(float("nan"))
                sm.spmma_grouped(blobs, Bs, Cs, m, n, k, batch=b)
                torch.cuda.synchronize()
                got = [C.view(torch.int16).clone() for C in Cs]
                if ref is None:
                    ref = got
                elif not all(torch.equal(x, y) for x, y in zip(ref, got)):
                    print(f"   {m}x{n}x{k} x{cnt} [{name}] C DIFFERS from [base]", flush=True)
            times.setdefault(name, []).append(sm.graph_time_ms(lambda: sm.spmma_grouped(blobs, Bs, Cs, m, n, k, batch=b), iters=4) * 1e3)
    print(f"{m}x{n}x{k} b={b} x{cnt}  roof {by / 8e6:6.1f} us | " + "  ".join(f"{name}: {min(ts):7.1f} us ({by / min(ts) / 1e6:5.2f} TB/s)" for name, ts in times.items()), flush=True)
    for name, ts in times.items():
        tot[name] += min(ts)
    del blobs, Bs, Cs, ref
for key in KEYS:
    os.environ.pop(key, None)
print("# sums (us): " + "  ".join(f"{k_}: {v:.0f}" for k_, v in tot.items()))
