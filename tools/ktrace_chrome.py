#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> Chrome trace JSON (chrome://tracing, ui.perfetto.dev): one track per hardware queue,
one slice per kernel dispatch.  The counterpart of the chrome trace the reference exports from torch.profiler
(datasets/get_shapes.py:75-85, prof.export_chrome_trace) for this build's kernels; ROCTX ranges of the C++ operators
(include/sparsify.me/util/trace.hxx) come through `rocprofv3 --marker-trace` the same way.
usage: ktrace_chrome.py <rocprofv3 output dir> <out.json>"""
import csv
import glob
import json
import sys


def main():
    d, out = sys.argv[1:3]
    events, t0 = [], None
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            t0 = s if t0 is None else min(t0, s)
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            events.append({"name": name, "ph": "X", "pid": int("".join(ch for ch in str(r.get("Agent_Id", "0")) if ch.isdigit()) or 0), "tid": int(r["Queue_Id"]),
                           "ts": s, "dur": e - s,
                           "args": {"grid": [int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])],
                                    "workgroup": int(r["Workgroup_Size_X"]), "lds": int(r["LDS_Block_Size"]),
                                    "vgprs": int(r["VGPR_Count"]), "dispatch": int(r["Dispatch_Id"])}})
    for ev in events:  # ns since the first dispatch -> us
        ev["ts"] = (ev["ts"] - t0) / 1e3
        ev["dur"] = ev["dur"] / 1e3
    json.dump({"traceEvents": events, "displayTimeUnit": "ns"}, open(out, "w"))
    print(f"{len(events)} kernel slices -> {out}")


if __name__ == "__main__":
    main()
