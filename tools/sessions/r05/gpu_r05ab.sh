#!/bin/bash
# per-kernel durations of the sparse-instruction COO call
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05ab}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 tools/coo_profile.py 12544,64,576 196,512,4608 3136,128,1152 > gpurun_out/${tag}_profile.txt 2>&1; rc=$?; tail -5 gpurun_out/${tag}_profile.txt
[ $rc = 0 ] || exit 1
f=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/${tag}_kernels.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size"] if "Grid_Size" in r else "") for r in rows]
# print the last call of each shape: find groups ending in spmm_coo_smfmac_kernel
out = []
for i, (n_, d, g) in enumerate(seq):
    if "spmm_coo_smfmac_kernel" in n_:
        j = i
        grp = []
        while j >= 0 and len(grp) < 5:
            grp.append(seq[j]); j -= 1
            if j >= 0 and "spmm_coo_smfmac_kernel" in seq[j][0]: break
        out.append(list(reversed(grp)))
for grp in out[3::9] + out[-1:]:
    print(" | ".join("%s %.1f" % (n_.replace("void sm::", "")[:28], d) for n_, d, g in grp))
PY
