#!/usr/bin/env python3
"""Yardstick only (nothing in the library, tests or bench uses it): what a plain device-to-device copy and a plain
read reach on this part, next to this library's streaming kernels.  tools/copy_yardstick.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device("cuda", 0)
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.float16, device=dev); sm.fill_uniform(x, 1, 0.0, 1.0)
    y = torch.empty_like(x)
    t = sm.graph_time_ms(lambda: y.copy_(x), iters=5, replays=3)
    print(f"copy   {mb:5d} MiB: {t * 1e3:8.1f} us  {2 * n * 2 / t / 1e9:7.2f} TB/s (read + write)")
    t = sm.graph_time_ms(lambda: torch.sum(x, dtype=torch.float32), iters=5, replays=3)
    print(f"reduce {mb:5d} MiB: {t * 1e3:8.1f} us  {n * 2 / t / 1e9:7.2f} TB/s (read)")
    t = sm.graph_time_ms(lambda: y.fill_(1.5), iters=5, replays=3)
    print(f"fill   {mb:5d} MiB: {t * 1e3:8.1f} us  {n * 2 / t / 1e9:7.2f} TB/s (write)")
    m, k = n // 4096, 4096
    t = sm.graph_time_ms(lambda: sm.prune24(x, y, m, k, k, sm.PRUNE_STRIP), iters=5, replays=3)
    print(f"sm_prune24 STRIP (out of place) {mb:5d} MiB: {t * 1e3:8.1f} us  {2 * n * 2 / t / 1e9:7.2f} TB/s")
    blob = torch.empty(sm.compress24_size(m, k, 2, 1), dtype=torch.uint8, device=dev)
    t = sm.graph_time_ms(lambda: sm.compress24(x, m, k, k, 1, m * k, blob), iters=5, replays=3)
    print(f"sm_compress24                   {mb:5d} MiB: {t * 1e3:8.1f} us  {(n * 2 + blob.numel()) / t / 1e9:7.2f} TB/s")
