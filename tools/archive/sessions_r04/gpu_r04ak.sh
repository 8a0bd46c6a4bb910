#!/bin/bash
# round 4, session ak: im2col staging with 2 / 4 narrow segments per wave instruction: parity, conv routes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "im2col or conv" > gpurun_out/r04ak_pytest.txt 2>&1; guard $? "pytest"; tail -4 gpurun_out/r04ak_pytest.txt
timeout -k 10 400 python tools/conv_routes.py > gpurun_out/r04ak_conv_routes.txt 2> gpurun_out/r04ak_conv_routes.err; guard $? "conv routes"
cat gpurun_out/r04ak_conv_routes.txt
