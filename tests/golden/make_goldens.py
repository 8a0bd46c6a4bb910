#!/usr/bin/env python3
"""Generates tests/golden/resnet50_goldens.json with the ORACLE (oracle/sm_oracle.c) from seeded inputs:
for one batch (b = 1) of each of the 17 unique ResNet-50 (m, k) operand shapes, fp16 and fp32:
sha256 of the STRIP-pruned matrix, the TILE-pruned matrix and the compressed blob, plus a few sampled
entries; and for BASELINE config 1 (512 x 512 x 512 fp32: prune-to-2:4 + dense GEMM of the pruned A) sampled
entries of C.  Inputs: numpy Generator(PCG64(seed)).random(float32) in [0, 1) -- the distribution the reference's
drivers use (examples/spmma.cu:51-55) -- rounded to the element type.

The reference itself has no golden vectors and cannot run here (section 8c), so these
fixtures pin THIS BUILD's frozen semantics over time and let the GPU tests check full-size layers without
running the oracle.  Run from the repo root:  python tests/golden/make_goldens.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

SHAPES_MK = sorted({(12544, 147), (12544, 64), (12544, 576), (12544, 256), (3136, 1152), (3136, 128), (3136, 512),
                    (784, 2304), (784, 256), (784, 1024), (196, 4608), (196, 512), (196, 2048)})
DTYPES = {"f16": np.float16, "f32": np.float32}


def gen(seed, count, dtype):
    return np.random.Generator(np.random.PCG64(seed)).random(count, dtype=np.float32).astype(dtype)


def bits(a):
    return a.view({2: np.uint16, 4: np.uint32}[a.dtype.itemsize])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    orc = ge.load_oracle()
    out = {"_generator": "tests/golden/make_goldens.py", "_inputs": "PCG64(seed).random(float32) in [0,1), cast to dtype",
           "operands": [], "config1": {}}
    for (m, k) in SHAPES_MK:
        for name, dt in DTYPES.items():
            seed = 1000003 * m + 101 * k + (0 if name == "f16" else 7)
            A = gen(seed, m * k, dt)
            ps = orc.prune24(bits(A), m, k, k, orc.STRIP)
            pt = orc.prune24(bits(A), m, k, k, orc.TILE)
            blob = orc.compress24(bits(A), m, k, k)
            idx = [0, 1, 2, 3, (m * k) // 2, m * k - 4, m * k - 3, m * k - 2, m * k - 1]
            out["operands"].append({"m": m, "k": k, "dtype": name, "seed": seed, "input_sha256": sha(bits(A)),
                                    "strip_sha256": sha(ps), "tile_sha256": sha(pt), "blob_sha256": sha(blob),
                                    "blob_bytes": int(blob.size), "sample_idx": idx,
                                    "strip_sample_bits": [int(ps[i]) for i in idx],
                                    "tile_sample_bits": [int(pt[i]) for i in idx],
                                    "valid_after_strip": orc.prune24_check(ps, m, k, k),
                                    "valid_after_tile": orc.prune24_check(pt, m, k, k),
                                    "valid_dense": orc.prune24_check(bits(A), m, k, k)})
    # config 1: single 512 x 512 x 512 fp32 layer: prune to 2:4 (STRIP) + dense GEMM of the pruned A, fp64 accumulate
    m = n = k = 512
    A, B = gen(0x5EED, m * k, np.float32), gen(0x5EED + 1, k * n, np.float32)
    P = orc.prune24(bits(A), m, k, k, orc.STRIP).view(np.float32)
    C = np.zeros(m * n, dtype=np.float32)
    orc.gemm_rowmajor(P, B, C, m, n, k)
    C2 = np.zeros(m * n, dtype=np.float32)
    orc.spmma(orc.compress24(bits(A), m, k, k), B, C2, m, n, k)
    assert np.array_equal(C, C2)
    idx = [0, 1, 511, 512, 131071, 131072, 262143]
    out["config1"] = {"m": m, "n": n, "k": k, "dtype": "f32", "seedA": 0x5EED, "seedB": 0x5EED + 1,
                      "pruned_sha256": sha(bits(P)), "sample_idx": idx, "C_sample": [float(C[i]) for i in idx],
                      "C_sum": float(C.astype(np.float64).sum())}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resnet50_goldens.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, len(out["operands"]), "operand fixtures")


if __name__ == "__main__":
    main()
