#!/bin/bash
# round 6, session f: the prune + flag pass without the blob's re-selection (prune_compress_kernel<..., BLOB = false>): parity, API table, bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "prune" > gpurun_out/r06f_pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r06f_pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/api_path_table.py > gpurun_out/r06f_api_path.txt 2> gpurun_out/r06f_api_path.err; echo "api table rc=$?"; cat gpurun_out/r06f_api_path.txt
bash tools/gpu_bench_only.sh r06f
