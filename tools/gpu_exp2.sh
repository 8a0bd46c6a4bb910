#!/bin/bash
set -o pipefail
tag=${1:-r03e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 300 -k "headers or one_pass_f32" > gpurun_out/${tag}_pytest.log 2>&1; echo pytest rc=$?; tail -3 gpurun_out/${tag}_pytest.log
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
step s8_f512 $B --streams 8 --fused-max-n 512
step s6_f512 $B --streams 6 --fused-max-n 512
step s12_f512 $B --streams 12 --fused-max-n 512
step s4_f512 $B --streams 4 --fused-max-n 512
step s8_f512_b $B --streams 8 --fused-max-n 512
step s8_f256 $B --streams 8
step s16_f512 $B --streams 16 --fused-max-n 512
step s8_f512_nogroup $B --streams 8 --fused-max-n 512 --group off
bash tools/pmc_families.sh ${tag}
