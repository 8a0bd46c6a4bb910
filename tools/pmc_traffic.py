#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE in one, WRITE_SIZE in the other; they do not fit one pass:
MI355X_MICROARCH.md 'rocprofv3 PMC slots') into HBM bytes per launch per kernel.

Correction applied as the guide's HBM section prescribes for gfx950: FETCH_SIZE reports exactly half the
bytes of a wide coalesced streaming read (16 B per lane, plain loads and LDS-DMA alike), so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane stores.  Both counters are in KiB.

usage: pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json> [library.so]
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


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                name = name.split("<")[0].split("::")[-1]
                acc[name].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def durations(d):
    """average End-Start (ns) per kernel over the dispatches of a pass (the PMC rows carry the timestamps)"""
    acc, seen = collections.defaultdict(list), set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (f, r["Dispatch_Id"])
            if key in seen:
                continue
            seen.add(key)
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip().split("<")[0].split("::")[-1]
            acc[name].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    dur = durations(fdir)
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        res[k] = {"launches_profiled": max(nf, nw), "FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w,
                  "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                  "correction": "2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B) + WRITE_SIZE, KiB -> B"}
        if dur.get(k):
            res[k]["avg_duration_us_in_pmc_pass"] = dur[k] / 1e3
            res[k]["hbm_GBs"] = res[k]["hbm_bytes_per_launch"] / dur[k]
    res_out = dict(res)
    res_out["_library"] = library_tag(sys.argv[4] if len(sys.argv) > 4 else None)
    json.dump(res_out, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:32s} launches {v['launches_profiled']:5d}  hbm {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch"
              + (f"  {v['hbm_GBs']:8.1f} GB/s" if "hbm_GBs" in v else ""))


if __name__ == "__main__":
    main()
