#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected.  usage: pmc_generic.py <pass_dir>"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[-60:]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    print(n, {k: round(sum(v) / len(v)) for k, v in sorted(d.items())}, "launches", max(len(v) for v in d.values()))
