#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05af}
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "coo_smfmac or coo_fast" > gpurun_out/${tag}_tests.txt 2>&1; rc=$?; tail -3 gpurun_out/${tag}_tests.txt
if [ $rc != 0 ]; then echo "tests rc=$rc"; exit 1; fi
timeout -k 10 300 python tools/coo_config5.py > gpurun_out/${tag}_config5.txt 2> gpurun_out/${tag}_config5.err; rc=$?; cat gpurun_out/${tag}_config5.txt
[ $rc = 0 ] || exit 1
export SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so
for w in 4 8; do for ab in 0 1 2 4 7; do
echo "== SM_COO_WAVES=$w SM_COO_ABLATE=$ab" | tee -a gpurun_out/${tag}_ablate.txt
SM_COO_WAVES=$w SM_COO_ABLATE=$ab timeout -k 10 200 python tools/coo_profile.py 12544,64,576 196,512,4608 3136,128,1152 2>/dev/null | tee -a gpurun_out/${tag}_ablate.txt || exit 1
done; done
