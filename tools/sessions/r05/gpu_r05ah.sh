#!/bin/bash
# config 5, both matrix-core forms on the same box (tuning library: SM_COO_SMFMAC = 2 wherever the sparse-instruction form can run, 0 never)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05ah}
export SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so
for f in 0 2 0; do
echo "== SM_COO_SMFMAC=$f" | tee -a gpurun_out/${tag}_forms.txt
SM_COO_SMFMAC=$f timeout -k 10 300 python tools/coo_config5.py 2>/dev/null | tee -a gpurun_out/${tag}_forms.txt || exit 1
done
