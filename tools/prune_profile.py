#!/usr/bin/env python3
"""Runs the prune-step kernels (STRIP prune, TILE prune, check, compress, one-pass prune+check+compress) once per unique ResNet-50 shape at b = 32
(fp16), for rocprofv3 passes: tools/prune_profile.py [reps].  Used by tools/gpu_round.sh to report the prune step's
HBM GB/s from the PMC counters (north_star: "rocprof-reported HBM GB/s for the prune step")."""
import csv
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda", 0)
rows = [tuple(int(x) for x in r[:3]) for r in list(csv.reader(open(os.path.join(ROOT, "datasets", "resnet50.csv"))))[1:] if r]
b = 32
valid = torch.zeros(1, dtype=torch.int32, device=dev)
for (m, n, k) in sorted(set(rows)):
    A = torch.empty(b * m * k, dtype=torch.float16, device=dev)
    sm.fill_uniform(A, 1, -1.0, 1.0)
    O = torch.empty_like(A)
    blob = torch.empty(sm.compress24_size(m, k, 2, b), dtype=torch.uint8, device=dev)
    for _ in range(reps):
        sm.prune24(A, O, b * m, k, k, 1)   # STRIP
        sm.prune24(A, O, b * m, k, k, 0)   # TILE
        sm.prune24_check(O, b * m, k, k, valid)
        sm.compress24(A, m, k, k, b, m * k, blob)
        sm.prune24_compress24(A, O, m, k, k, b, m * k, blob, valid, sm.PRUNE_TILE)    # one pass (spmma()'s sequence)
    torch.cuda.synchronize()
