#!/bin/bash
set -o pipefail
tag=${1:-r03q}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 300 -x -k "coo_fast" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; echo pytest rc=$rc; tail -3 gpurun_out/${tag}_pytest.log
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
step base $B
step model $B --cost model
step small $B --item-order small-first
step model_small $B --cost model --item-order small-first
step base_b $B
step model_b $B --cost model
step small_b $B --item-order small-first
step s6 $B --streams 6
step s5 $B --streams 5
timeout -k 10 200 python3 - <<'PY'
import sys
sys.path.insert(0, '.')
import torch, bench
import __graft_entry__ as ge
sm = ge.load_package()
r = bench.config5_stage(sm, torch, torch.device('cuda', 0))
print('coo fast', [round(s['ms_fast_form'] * 1e3, 1) for s in r['shapes']], 'packed', [round(s['ms'] * 1e3, 1) for s in r['shapes']])
PY
