#!/bin/bash
# round 4, session y: span form of the fp32 split kernel: parity re-run (the test's exact twin for ragged k fixed)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "f32_split or cpp or drivers" > gpurun_out/r04y_pytest.txt 2>&1; guard $? "pytest"; tail -5 gpurun_out/r04y_pytest.txt
