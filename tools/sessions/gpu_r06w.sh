#!/bin/bash
# round 6, last check: smoke() + the whole GPU suite (incl. the bench-line test) as the driver will run them
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06w_smoke.log 2>&1; rc=$?; tail -2 gpurun_out/r06w_smoke.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06w_pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r06w_pytest.log; exit $rc
