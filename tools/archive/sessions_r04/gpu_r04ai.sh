#!/bin/bash
# round 4, session ai: flag raising with a relaxed read first; check / prune rates on valid and invalid operands; parity of every flag test
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "prune or check or goldens or i8 or one_pass or one_kernel" > gpurun_out/r04ai_pytest.txt 2>&1; guard $? "pytest"; tail -4 gpurun_out/r04ai_pytest.txt
timeout -k 10 300 python tools/prune_rates.py > gpurun_out/r04ai_prune_rates.txt 2> gpurun_out/r04ai_prune_rates.err; guard $? "rates"
cat gpurun_out/r04ai_prune_rates.txt
