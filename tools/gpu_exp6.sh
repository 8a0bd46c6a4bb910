#!/bin/bash
set -o pipefail
tag=${1:-r03i}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -x -k "fused or grouped" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; echo pytest rc=$rc; tail -5 gpurun_out/${tag}_pytest.log
if [ "$rc" != 0 ]; then exit 1; fi
step() { label=$1; shift
  out=$(env "$@" 2>gpurun_out/${tag}_${label}.err); rc=$?
  echo "$out" > gpurun_out/${tag}_${label}.json
  python3 -c "
import json
try:
    d=json.loads(open('gpurun_out/${tag}_${label}.json').read().strip().splitlines()[-1]); print('$label', 'ms_per_step', round(d['ms_per_step'],4), d['config']['path'][:60])
except Exception as e: print('$label', 'failed', e)
"
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$label hit its limit; stopping"; exit 1; fi
}
B="timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline"
step span $B
step nospan $B --no-span
step span_b $B
step nospan_b $B --no-span
python3 tools/time_fused.py 12544 64 147 32 1
