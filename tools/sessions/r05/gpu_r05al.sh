#!/bin/bash
# the producer / consumer form of the sparse-instruction COO kernel: tests, both forms wherever they can run (tuning library), ablations
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05al}
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "coo_smfmac or coo_fast" > gpurun_out/${tag}_tests.txt 2>&1; rc=$?; tail -3 gpurun_out/${tag}_tests.txt
if [ $rc != 0 ]; then echo "tests rc=$rc"; exit 1; fi
export SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so
for pc in 1 0; do
echo "== SM_COO_SMFMAC=2 SM_COO_PC=$pc" | tee -a gpurun_out/${tag}_forms.txt
SM_COO_SMFMAC=2 SM_COO_PC=$pc timeout -k 10 300 python tools/coo_config5.py 2>/dev/null | tee -a gpurun_out/${tag}_forms.txt || exit 1
done
for ab in 1 2 4 7; do
echo "== SM_COO_SMFMAC=2 SM_COO_ABLATE=$ab" | tee -a gpurun_out/${tag}_ablate.txt
SM_COO_SMFMAC=2 SM_COO_ABLATE=$ab timeout -k 10 200 python tools/coo_profile.py 12544,64,576 196,512,4608 3136,128,1152 2>/dev/null | tee -a gpurun_out/${tag}_ablate.txt || exit 1
done
