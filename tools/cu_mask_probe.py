#!/usr/bin/env python3
"""Does partitioning the CUs between the bandwidth-bound launches (direct family) and the few-tile launches (big / wide / A-stationary / span) of the
ResNet-50 step shorten it?  Streams with CU masks (hipExtStreamCreateWithCUMask), eager launches, one MI355X.
usage: python tools/cu_mask_probe.py [steps]"""
import ctypes, os, sys, time, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
import bench
sm = ge.load_package()
dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10

def masked_stream(pred):
    words = (ctypes.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32); n += 1
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return torch.cuda.ExternalStream(s.value, device=dev), n

shapes = bench.read_shapes(bench.table_path("resnet50"))
cnt = collections.Counter(shapes)
groups = []
for (m, n, k, b), c in cnt.items():
    fam = bench.fused_variant(n, k, m, b, min(c, 8))
    As, Bs, Cs = [], [], []
    for i in range(c):
        A = torch.empty(b * m * k, dtype=torch.float16, device=dev); sm.fill_uniform(A, 1 + i, -1.0, 1.0)
        B = torch.empty(k * n, dtype=torch.float16, device=dev); sm.fill_uniform(B, 20 + i, -1.0, 1.0)
        As.append(A); Bs.append(B); Cs.append(torch.empty(b * m * n, dtype=torch.float16, device=dev))
    bytes_ = c * (b * m * k * 2 + k * n * 2 + b * m * n * 2)
    groups.append(dict(shape=(m, n, k, b), fam=fam, As=As, Bs=Bs, Cs=Cs, bytes=bytes_))
groups.sort(key=lambda g: -g["bytes"])
bw = [g for g in groups if g["fam"] in ("direct",)]
ft = [g for g in groups if g["fam"] not in ("direct",)]
print("direct launches:", [g["shape"][:3] for g in bw], " few-tile launches:", [(g["fam"],) + g["shape"][:3] for g in ft], flush=True)

def launch(g):
    sm.spmma_fused_grouped(g["As"], g["Bs"], g["Cs"], *g["shape"][:3], batch=g["shape"][3])

def run(assign, label):
    """assign: list of (stream, [groups]); launches are issued round-robin over the streams so that no stream's queue starves"""
    def one_step():
        qs = [(s, list(gs)) for s, gs in assign]
        while any(q for _, q in qs):
            for s, q in qs:
                if q:
                    with torch.cuda.stream(s):
                        launch(q.pop(0))
    for _ in range(3): one_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): one_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print(f"{label:58s} {dt:7.3f} ms per step", flush=True)
    return dt

plain = [torch.cuda.Stream(device=dev) for _ in range(8)]
run([(plain[0], bw + ft)], "one stream, all launches")
run([(plain[0], bw), (plain[1], ft)], "two plain streams (direct | few-tile)")
rr = [(plain[i], (bw + ft)[i::8]) for i in range(8)]
run(rr, "eight plain streams, round robin by bytes")
for frac_ft in (64, 96, 128, 160):
    # CU i of the mask: the few-tile stream takes those with (i // 8) % 4 in a set sized to frac_ft / 256 (8-CU granules, balanced over XCDs under either bit order)
    gran = frac_ft // 8
    pick = set(range(0, 32, 32 // gran)) if 32 % gran == 0 else set(range(gran))
    pick = set(sorted(pick)[:gran])
    s_ft, n_ft = masked_stream(lambda i: (i // 8) in pick)
    s_bw, n_bw = masked_stream(lambda i: (i // 8) not in pick)
    run([(s_bw, bw), (s_ft, ft)], f"masked: direct on {n_bw} CUs | few-tile on {n_ft} CUs")
    # few-tile spread over two masked streams sharing the same CUs
    s_ft2, _ = masked_stream(lambda i: (i // 8) in pick)
    run([(s_bw, bw), (s_ft, ft[0::2]), (s_ft2, ft[1::2])], f"  same, few-tile launches on two streams of those CUs")
