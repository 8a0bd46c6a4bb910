#!/bin/bash
# round 5, session q: the non-temporal hint on the direct kernel's A loads, per shape and in the step (tuning library)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so
timeout -k 10 300 python tools/nt_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05q_nt_ab.txt; cat gpurun_out/r05q_nt_ab.txt
for nt in 1 0 1 0; do
SM_DIRECT_NT=$nt timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r05q_bench_nt$nt.json 2> gpurun_out/r05q_bench_nt$nt.err; echo "bench nt=$nt rc=$?"
python3 -c "
import json
d=json.load(open('gpurun_out/r05q_bench_nt$nt.json')); print('SM_DIRECT_NT=$nt: ms_per_step', round(d['ms_per_step'],4), 'verified', d.get('verified'))"
done
