#!/usr/bin/env python3
"""Concurrency profile of one bench step from a rocprofv3 --kernel-trace CSV: per queue first start / last end / busy
time, and the time spent with 0, 1, 2, ... kernels active.  tools/ktrace_step.py <dir> [kernels_per_step]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fill" not in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 56
last = rows[-n:]
t0 = min(int(r["Start_Timestamp"]) for r in last); t1 = max(int(r["End_Timestamp"]) for r in last)
print(f"kernels {len(last)}  span {(t1 - t0) / 1e3:.1f} us  sum of durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e3:.1f} us")
ev = []
for r in last:
    ev += [(int(r["Start_Timestamp"]), 1), (int(r["End_Timestamp"]), -1)]
ev.sort()
act, prev, hist = 0, t0, {}
for t, d in ev:
    hist[act] = hist.get(act, 0) + (t - prev); prev = t; act += d
print("time with k kernels active (us):", {k: round(v / 1e3, 1) for k, v in sorted(hist.items())})
q = collections.defaultdict(list)
for r in last:
    q[r["Queue_Id"]].append(r)
for k, v in sorted(q.items()):
    s = min(int(r["Start_Timestamp"]) for r in v) - t0; e = max(int(r["End_Timestamp"]) for r in v) - t0
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v)
    print(f"queue {k}: {len(v):3d} kernels  first start {s / 1e3:8.1f}  last end {e / 1e3:8.1f}  busy {busy / 1e3:8.1f} us")
