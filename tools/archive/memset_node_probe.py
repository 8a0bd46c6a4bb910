#!/usr/bin/env python3
"""What a 4-byte hipMemsetAsync costs inside a replayed hipGraph (the flag reset of sm_prune24_check / sm_prune24_compress24 /
sm_prune24_spmma), against a one-block kernel: graph_time_ms of (a) sm_prune24_check on an 8-element matrix (memset node + tiny kernel),
(b) a tiny fill kernel alone, (c) check on the large operand.  usage: python tools/memset_node_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
tiny = torch.zeros(8, dtype=torch.float16, device=dev)
v = torch.zeros(1, dtype=torch.int32, device=dev)
big = torch.empty(32 * 12544 * 576, dtype=torch.float16, device=dev); sm.fill_uniform(big, 3, -1.0, 1.0)
t = lambda fn, it=20: min(sm.graph_time_ms(fn, iters=it) for _ in range(3)) * 1e3
print("check on 8 elements (memset node + one-block kernel): %.1f us" % t(lambda: sm.prune24_check(tiny, 1, 8, 8, v)))
print("fill kernel on 8 elements (one-block kernel alone):   %.1f us" % t(lambda: sm.fill_uniform(tiny, 1, 0.0, 1.0)))
print("check on 12544 x 576 x 32 fp16 (462 MB):              %.1f us" % t(lambda: sm.prune24_check(big, 32 * 12544, 576, 576, v), 6))
