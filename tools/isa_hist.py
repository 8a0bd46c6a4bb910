#!/usr/bin/env python3
"""isa_hist.py <file.s> <kernel-name-substring>: instruction histogram of one kernel of a -save-temps assembly file."""
import collections
import sys

s = open(sys.argv[1]).read()
names = [l.split(":")[0] for l in s.splitlines() if sys.argv[2] in l and l.startswith("_") and ":" in l]
for name in names:
    i = s.index("\n" + name + ":")
    j = s.index(".Lfunc_end", i)
    cnt = collections.Counter()
    for l in s[i:j].splitlines()[1:]:
        l = l.strip()
        if not l or l.startswith((".", ";")) or l.split(";")[0].strip().endswith(":"):
            continue
        cnt[l.split()[0]] += 1
    tot = sum(cnt.values())
    valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
    print(f"{name}: {tot} instructions, {valu} VALU")
    print("  " + "  ".join(f"{k} {v}" for k, v in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 16)))
