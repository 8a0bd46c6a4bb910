#!/usr/bin/env python3
"""pmc_sum.py <rocprofv3 output dir> [kernel-substring]: per kernel, the sum of every collected counter over its dispatches
divided by the number of dispatches (counter_collection.csv of a `rocprofv3 --kernel-trace --pmc ...` run)."""
import collections
import csv
import glob
import sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sub and sub not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    n = len(disp[k])
    print(k[-60:], "dispatches", n, {name: round(v / n) for name, v in sorted(c.items())})
