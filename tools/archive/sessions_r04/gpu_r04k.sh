#!/bin/bash
# round 4, session k: conv kernel levers one by one (16-byte patch DMAs, small plan + occupancy hint, 256-column tiles)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$PWD/sparsify.me_amd
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "conv" > gpurun_out/r04k_pytest_conv.txt 2>&1; guard $? "pytest conv"; tail -2 gpurun_out/r04k_pytest_conv.txt
rm -f gpurun_out/r04k_conv_levers.txt
for v in "0 0 0" "1 0 0" "1 1 0" "1 1 1" "0 1 1"; do
  set -- $v
  echo "== SM_CONV_V16=$1 SM_CONV_SMALL=$2 SM_CONV_BN256=$3" >> gpurun_out/r04k_conv_levers.txt
  CONV_PROBE_ONLY=implicit SM_CONV_V16=$1 SM_CONV_SMALL=$2 SM_CONV_BN256=$3 SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 300 python tools/conv_probe.py >> gpurun_out/r04k_conv_levers.txt 2>&1; guard $? "conv probe $v"
done
grep -v "amdgpu.ids" gpurun_out/r04k_conv_levers.txt | grep -v "^  Cin"
