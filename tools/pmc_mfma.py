#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from one rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE (rocprofv3's own derived metric MfmaUtil = busy / (GUI_ACTIVE x SIMDs) x 100):
the share of the kernel's cycles in which a SIMD's matrix pipe was executing, averaged over the 1024 SIMDs.
In the CSV both counters arrive summed over the 8 XCDs: SQ_VALU_MFMA_BUSY_CYCLES is then exactly
(matrix instructions issued) x (their cycles) -- 401 408 v_smfmac_f32_16x16x64_f16 x 16 cycles = 6 422 528 for
the 12544 x 256 x 64 layer at b = 32 -- and GRBM_GUI_ACTIVE is 8 x the kernel's duration in shader cycles, so the
per-XCD duration (rocprofv3's reduce(...,max)) is the sum / 8.

usage: pmc_mfma.py <pass_dir> <out.json> [library.so]
(the library's sha256 goes into the output under "_library": bench.py replays the file only for the library it was measured on)"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys


def library_tag(path=None):
    """sha256 (first 16 hex digits) of the library the profiled process loaded: SPARSIFYME_LIB or the in-tree product library"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = path or os.environ.get("SPARSIFYME_LIB") or os.path.join(root, "sparsify.me_amd", "libsparsifyme.so")
    try:
        with open(path, "rb") as fh:
            return {"sha256_16": hashlib.sha256(fh.read()).hexdigest()[:16], "lib_path": os.path.relpath(path, root)}
    except OSError:
        return {"sha256_16": None, "lib_path": path}

SIMDS = 256 * 4
XCDS = 8


def main():
    d, out = sys.argv[1:3]
    busy, act = collections.defaultdict(dict), collections.defaultdict(dict)
    names = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (f, r["Dispatch_Id"])
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").strip().split("<")[0].split("::")[-1]
            names[key] = n
            if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                busy[key] = busy.get(key, 0.0) + float(r["Counter_Value"])
            elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                act[key] = max(act.get(key, 0.0), float(r["Counter_Value"]))
    per = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for key, n in names.items():
        if key in busy and key in act and act[key] > 0:
            per[n][0] += busy[key]
            per[n][1] += act[key]
            per[n][2] += 1
    res = {n: {"launches_profiled": c, "mfma_busy_cycles": b, "gui_active_cycles": a,
               "mfma_util_percent": 100.0 * b / (a / XCDS * SIMDS)} for n, (b, a, c) in sorted(per.items()) if b > 0}
    res_out = dict(res)
    res_out["_library"] = library_tag(sys.argv[3] if len(sys.argv) > 3 else None)
    json.dump(res_out, open(out, "w"), indent=1)
    for n, v in res.items():
        print(f"{n:32s} launches {v['launches_profiled']:5d}  MfmaUtil {v['mfma_util_percent']:6.2f} %")


if __name__ == "__main__":
    main()
