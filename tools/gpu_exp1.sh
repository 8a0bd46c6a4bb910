#!/bin/bash
# round-3 experiment session: full GPU suite, then A/B runs of the timed step (no extras)
set -o pipefail
tag=${1:-r03d}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${tag}_pytest.log 2>&1; guard $? pytest; tail -5 gpurun_out/${tag}_pytest.log
step() { # label, env..., -- args
  label=$1; shift
  out=$(env "$@" 2>gpurun_out/${tag}_${label}.err); rc=$?
  echo "$out" > gpurun_out/${tag}_${label}.json
  python3 -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/${tag}_${label}.json').read().strip().splitlines()[-1]); print('$label', 'ms_per_step', round(d['ms_per_step'],4))
except Exception as e: print('$label', 'failed', e)
"
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$label hit its limit; stopping"; exit 1; fi
}
B="timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline"
step base1 $B
step nt0 SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so SM_DIRECT_NT=0 $B
step nt1 SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so SM_DIRECT_NT=1 $B
step nt0b SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so SM_DIRECT_NT=0 $B
step nt1b SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so SM_DIRECT_NT=1 $B
step s2 $B --streams 2
step s3 $B --streams 3
step s6 $B --streams 6
step s8 $B --streams 8
step fmn512 $B --fused-max-n 512
step base2 $B
