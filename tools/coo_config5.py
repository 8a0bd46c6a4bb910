#!/usr/bin/env python3
"""bench.py's config-5 stage alone (sparsifyme::batched::strided_coo over the ResNet-50 shapes, b = 32, 10 % dense A): one line per shape,
fast form (and which one) against the exact forms.  SM_COO_SMFMAC=0 with the tuning library gives the dense-MFMA pipeline for an A/B."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import importlib  # noqa: E402

import __graft_entry__ as ge
sm = ge.load_package()
out = bench.config5_stage(sm, torch, torch.device("cuda:0"))
print("# " + str(sm.version()))
print("%6s %5s %5s | %8s %8s | %8s %6s  %s" % ("m", "n", "k", "exact us", "rowptr", "fast us", "frac", "form"))
for r in out["shapes"]:
    print("%6d %5d %5d | %8.1f %8.1f | %8.1f %6.3f  %s flag=%s" % (r["m"], r["n"], r["k"], r["ms_exact"] * 1e3, r["ms_exact_rowptr_form"] * 1e3, r["ms"] * 1e3, r["frac"],
                                                             r["form"].split(" (")[0], r["range_flag"]))
fr = [r["frac"] for r in out["shapes"]]
print("# min frac %.3f  shapes at >= 0.25: %d of %d" % (min(fr), sum(f >= 0.25 for f in fr), len(fr)))
