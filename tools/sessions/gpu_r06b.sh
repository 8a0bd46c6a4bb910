#!/bin/bash
# round 6, session b: the whole GPU suite after the API-path / ADVICE / parity-hygiene changes, then the default bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06b_pytest.log 2>&1; rc=$?; tail -15 gpurun_out/r06b_pytest.log; [ $rc -ne 0 ] && exit $rc
bash tools/gpu_bench_only.sh r06b
