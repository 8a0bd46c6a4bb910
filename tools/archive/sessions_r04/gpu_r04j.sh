#!/bin/bash
# round 4, session j: implicit-GEMM conv kernel with 16-byte patch DMAs + occupancy hints (levers of DESIGN.md 4.4), A/B against the 4-byte form
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$PWD/sparsify.me_amd
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "conv" > gpurun_out/r04j_pytest_conv.txt 2>&1; guard $? "pytest conv"; tail -2 gpurun_out/r04j_pytest_conv.txt
for v in 0 1; do
  echo "== SM_CONV_V16=$v" >> gpurun_out/r04j_conv_probe.txt
  CONV_PROBE_ONLY=implicit SM_CONV_V16=$v SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 300 python tools/conv_probe.py >> gpurun_out/r04j_conv_probe.txt 2>&1; guard $? "conv probe v16=$v"
done
timeout -k 10 300 python tools/conv_probe.py >> gpurun_out/r04j_conv_probe.txt 2>&1; guard $? "conv probe product"
cat gpurun_out/r04j_conv_probe.txt | grep -v amdgpu.ids
