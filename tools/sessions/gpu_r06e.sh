#!/bin/bash
# round 6, session e: the span form of the TILE prune (ragged rows) -- prune / prune_compress / prune_spmma parity, then the API table
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prune or model_zoo or smoke" > gpurun_out/r06e_pytest.log 2>&1; rc=$?; tail -6 gpurun_out/r06e_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/api_path_table.py > gpurun_out/r06e_api_path.txt 2> gpurun_out/r06e_api_path.err; echo "api table rc=$?"; head -4 gpurun_out/r06e_api_path.txt; tail -1 gpurun_out/r06e_api_path.txt
