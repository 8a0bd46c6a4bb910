#!/usr/bin/env python3
"""Generates tests/golden/ext_goldens.json with the ORACLE from seeded inputs, for the extensions of the path
(bfloat16, int8, im2col): sha256 of pruned matrices, blobs, int8 products and im2col operands on a few ResNet-50
operand shapes.  Like resnet50_goldens.json these pin THIS BUILD's frozen semantics over time (the reference has no
golden vectors and cannot run here).  Run from the repo root:  python tests/golden/make_ext_goldens.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def bf16_bits(seed, count):
    """uniform [-1, 1) rounded to bfloat16 (round to nearest even), as uint16 bit patterns"""
    x = (rng(seed).random(count, dtype=np.float32) * 2 - 1).view(np.uint32)
    return ((x + 0x7fff + ((x >> 16) & 1)) >> 16).astype(np.uint16)


def i8_values(seed, count):
    return rng(seed).integers(-128, 128, count).astype(np.int8)


def main():
    orc = ge.load_oracle()
    out = {"_generator": "tests/golden/make_ext_goldens.py", "bf16": [], "i8": [], "im2col": []}
    for (m, k) in [(3136, 128), (784, 256), (196, 512), (130, 147)]:
        seed = 7000003 * m + 11 * k
        A = bf16_bits(seed, m * k)
        out["bf16"].append({"m": m, "k": k, "seed": seed, "input_sha256": sha(A),
                            "strip_sha256": sha(orc.prune24(A, m, k, k, orc.STRIP, bf16=True)),
                            "tile_sha256": sha(orc.prune24(A, m, k, k, orc.TILE, bf16=True)),
                            "blob_sha256": sha(orc.compress24(A, m, k, k))})
    for (m, n, k) in [(3136, 64, 128), (784, 72, 256), (196, 40, 576), (130, 16, 64)]:
        seed = 9000011 * m + 13 * k + n
        A, B = i8_values(seed, m * k), i8_values(seed + 1, n * k)
        blob = orc.compress24(A.view(np.uint8), m, k, k)
        C = np.zeros(m * n, dtype=np.int32)
        orc.spmma_i8(blob, B, C, m, n, k)
        out["i8"].append({"m": m, "n": n, "k": k, "seed": seed, "input_sha256": sha(A),
                          "strip_sha256": sha(orc.prune24(A.view(np.uint8), m, k, k, orc.STRIP)),
                          "tile_sha256": sha(orc.prune24(A.view(np.uint8), m, k, k, orc.TILE)),
                          "blob_sha256": sha(blob), "product_sha256": sha(C),
                          "requant_sha256": sha(orc.requant_i8(C, 2.0 ** -9)), "requant_scale": 2.0 ** -9})
    for cfg in [(2, 3, 56, 56, 7, 7, 2, 3, 1), (1, 64, 28, 28, 3, 3, 1, 1, 1), (2, 128, 14, 14, 1, 1, 1, 0, 1), (1, 16, 20, 33, 3, 3, 2, 1, 2)]:
        N, C_, H, W, kh, kw, s, p, d = cfg
        seed = 5000017 + sum((i + 1) * v for i, v in enumerate(cfg))
        X = (rng(seed).random(N * C_ * H * W, dtype=np.float32) * 2 - 1).astype(np.float16).view(np.uint16)
        A = orc.im2col(X, N, C_, H, W, kh, kw, s, p, d)
        L = orc.conv_out_size(H, kh, s, p, d) * orc.conv_out_size(W, kw, s, p, d)
        K = C_ * kh * kw
        out["im2col"].append({"cfg": list(cfg), "seed": seed, "input_sha256": sha(X), "L": L, "K": K, "operand_sha256": sha(A),
                              "blob_sha256": sha(orc.compress24(A, L, K, K, N))})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ext_goldens.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
