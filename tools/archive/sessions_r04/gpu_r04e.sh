#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "coo_config5 or coo_fast_config5" > gpurun_out/r04e_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04e_pytest.txt
timeout -k 10 600 python tools/sweep_grouped.py --table resnet50 > gpurun_out/r04e_sweep.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/r04e_sweep.txt
