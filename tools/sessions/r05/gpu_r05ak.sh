#!/bin/bash
# per-kernel averages of one sm_spmm_coo_f32_fast call, one shape per rocprofv3 run
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05ak}
for shp in 3136,128,1152 12544,64,576 196,512,4608 784,256,1024 12544,64,64; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_$shp -- python3 tools/coo_profile.py $shp > gpurun_out/${tag}_profile.txt 2>&1 || exit 1
f=$(find gpurun_out/${tag}_prof_$shp -name "*kernel_stats.csv" | head -1)
echo "== $shp" | tee -a gpurun_out/${tag}_kernels.txt
python3 - "$f" <<'PY' | tee -a gpurun_out/${tag}_kernels.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void sm::", "")[:70]
    if "fill_uniform" in n: continue
    print("  %-70s calls %4s avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
