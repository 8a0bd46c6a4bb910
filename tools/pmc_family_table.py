#!/usr/bin/env python3
"""Derived per-kernel figures from the counter passes of tools/pmc_families.sh (one representative
layer per kernel family, b = 32): which limit each family is at.  usage: pmc_family_table.py <dir prefix>... (pass dirs)"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            n = n.replace("sm::", "")
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            k = (f, r["Dispatch_Id"])
            if k not in seen:
                seen.add(k)
                dur[n].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
A = lambda n, c: (sum(acc[n][c]) / len(acc[n][c])) if acc[n].get(c) else None
shape = {"spmma_f16_fused_direct_kernel<64": "12544x64x576 x3", "spmma_f16_fused_direct_kernel<128": "3136x128x1152 x4", "spmma_f16_fused_wide_kernel": "784x256x2304 x6",
         "spmma_f16_fused_big_kernel<256, false, 3, 2, true": "784x256x1024 x5 (big)", "spmma_f16_fused_big_kernel<256, false, 3, 2, false": "196x512x4608 x3 (big)",
         "spmma_f16_fused_span_kernel": "12544x64x147", "spmma_f16_fused_astat_kernel": "784x1024x256 x6", "spmma_f16_pc_kernel": "196x512x4608 (staged 2:4 matmul)", "gemm_f16_dma_kernel": "784x256x2304 (dense)",
         "copy_bytes_kernel": "1 GiB copy", "compress_flat_kernel": "196x4608 compress"}
print("%-46s %-28s %8s %8s %9s %9s %9s %9s %9s %9s %9s" % ("kernel", "layer (b = 32)", "us", "CUs", "rd/clk/CU", "lat cyc", "lines/CU", "wait%", "istall%", "active%", "VALU%"))
for n in acc:
    if "fill_uniform" in n:
        continue
    lay = next((v for k, v in shape.items() if n.startswith(k)), "")
    gui = A(n, "GRBM_GUI_ACTIVE")
    cyc = gui / 8 if gui else None           # per-XCD shader cycles of the launch
    req, lat = A(n, "TCP_TCC_READ_REQ_sum"), A(n, "TCP_TCC_READ_REQ_LATENCY_sum")
    wc = A(n, "SQ_WAVE_CYCLES")
    waves = A(n, "SQ_WAVES")
    busy_cu = A(n, "SQ_BUSY_CU_CYCLES")
    cus = 256.0  # chip average (the few-tile launches keep 196 of the 256 CUs busy: their per-busy-CU figures are 1.3 x these)
    rate = req / cyc / cus if (req and cyc) else None
    latc = lat / req if (req and lat) else None
    infl = rate * latc if (rate and latc) else None
    f = lambda x, s="%9.1f": (s % x) if x is not None else "        -"
    print("%-46s %-28s %8.1f %8.0f %s %s %s %s %s %s %s" % (n[:46], lay, sum(dur[n]) / len(dur[n]), cus, f(rate, "%9.3f"), f(latc, "%9.0f"), f(infl),
          f(100 * A(n, "SQ_WAIT_ANY") / wc if wc else None), f(100 * A(n, "SQ_WAIT_INST_ANY") / wc if wc else None),
          f(100 * A(n, "SQ_ACTIVE_INST_ANY") / wc if wc else None), f(100 * A(n, "SQ_ACTIVE_INST_VALU") / wc if (wc and A(n, "SQ_ACTIVE_INST_VALU")) else None)))
print("""
us = launch duration under the counter pass; rd/clk/CU = TCP->TCC 128-byte read requests per shader cycle and CU (GRBM_GUI_ACTIVE / 8 XCDs);
lat cyc = TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ (average TCP->L2 read latency, hits and misses); lines/CU = their product = 128-byte reads
a CU has outstanding on average (Little); wait% = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked in s_waitcnt / s_barrier), istall% = SQ_WAIT_INST_ANY
(issue stalls), active% = SQ_ACTIVE_INST_ANY, VALU% = SQ_ACTIVE_INST_VALU, all as shares of wave residency.""")
