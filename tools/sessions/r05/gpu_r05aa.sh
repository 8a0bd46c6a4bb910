#!/bin/bash
# round 5: the COO product on the sparse matrix instruction + prepared planes of the fp32 split form: tests, then the config-5 table
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05aa}
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "coo_smfmac or coo_fast or prepared" > gpurun_out/${tag}_tests.txt 2>&1; rc=$?; tail -5 gpurun_out/${tag}_tests.txt
if [ $rc != 0 ]; then echo "tests rc=$rc"; exit 1; fi
timeout -k 10 300 python tools/coo_config5.py > gpurun_out/${tag}_config5.txt 2> gpurun_out/${tag}_config5.err; rc=$?; cat gpurun_out/${tag}_config5.txt; tail -3 gpurun_out/${tag}_config5.err
[ $rc = 0 ] || exit 1
