#!/bin/bash
# round 4, session l: new big-form test; step composition (streams / cost model) with the round-4 kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "big_form or conv or spmma_fused_f32" > gpurun_out/r04l_pytest.txt 2>&1; guard $? "pytest"; tail -2 gpurun_out/r04l_pytest.txt
for v in "--streams 8" "--streams 4" "--streams 6" "--streams 12" "--streams 16" "--streams 8 --cost model" "--streams 8 --item-order small-first" "--streams 8"; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline $v > gpurun_out/r04l_b.json 2> gpurun_out/r04l_b.err; guard $? "bench $v"
  python3 -c "
import json; d=json.load(open('gpurun_out/r04l_b.json')); print('  [$v] ms_per_step', round(d['ms_per_step'],4))"
done
