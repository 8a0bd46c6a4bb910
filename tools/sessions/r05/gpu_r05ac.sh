#!/bin/bash
# the COO product on the sparse matrix instruction: tests, config-5 table, kernel-level times, ablations (tuning library)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05ac}
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "coo_smfmac or coo_fast" > gpurun_out/${tag}_tests.txt 2>&1; rc=$?; tail -3 gpurun_out/${tag}_tests.txt
if [ $rc != 0 ]; then echo "tests rc=$rc"; exit 1; fi
timeout -k 10 300 python tools/coo_config5.py > gpurun_out/${tag}_config5.txt 2> gpurun_out/${tag}_config5.err; rc=$?; cat gpurun_out/${tag}_config5.txt
[ $rc = 0 ] || exit 1
export SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so
for ab in 0 1 2 4 6 7; do
echo "== SM_COO_ABLATE=$ab"
SM_COO_ABLATE=$ab timeout -k 10 200 python tools/coo_profile.py 12544,64,576 196,512,4608 3136,128,1152 12544,256,64 2>/dev/null | tee -a gpurun_out/${tag}_ablate.txt || exit 1
done
unset SPARSIFYME_LIB
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 tools/coo_profile.py 12544,64,576 196,512,4608 3136,128,1152 > gpurun_out/${tag}_profile.txt 2>&1 || exit 1
f=$(find gpurun_out/${tag}_prof -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/${tag}_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
out = []
for i, (n_, d) in enumerate(seq):
    if "spmm_coo_smfmac_kernel" in n_:
        out.append(seq[max(0, i - 4):i + 1])
for grp in out[3::9] + out[-1:]:
    print(" | ".join("%s %.1f" % (n_.replace("void sm::", "")[:28], d) for n_, d in grp))
PY
