#!/bin/bash
set -o pipefail
tag=${1:-r03l}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -x -k "fused or grouped" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; echo pytest rc=$rc; tail -5 gpurun_out/${tag}_pytest.log
if [ "$rc" != 0 ]; then exit 1; fi
T="SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so"
for v in 1 0; do
  env $T SM_FUSED_ASTATP=$v timeout -k 10 100 python3 tools/time_fused.py 3136 512 128 32 4 2>/dev/null | sed "s/^/astatp=$v /"
  env $T SM_FUSED_ASTATP=$v timeout -k 10 100 python3 tools/time_fused.py 784 1024 256 32 6 2>/dev/null | sed "s/^/astatp=$v /"
done
step() { label=$1; shift
  out=$(env "$@" 2>gpurun_out/${tag}_${label}.err); rc=$?
  echo "$out" > gpurun_out/${tag}_${label}.json
  python3 -c "
import json
try:
    d=json.loads(open('gpurun_out/${tag}_${label}.json').read().strip().splitlines()[-1]); print('$label', 'ms_per_step', round(d['ms_per_step'],4))
except Exception as e: print('$label', 'failed', e)
"
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$label hit its limit; stopping"; exit 1; fi
}
B="timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline"
step ap1 $T SM_FUSED_ASTATP=1 $B
step ap0 $T SM_FUSED_ASTATP=0 $B
step ap1b $T SM_FUSED_ASTATP=1 $B
step ap0b $T SM_FUSED_ASTATP=0 $B
step prod $B
