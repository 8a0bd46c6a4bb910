#!/bin/bash
set -o pipefail
tag=${1:-r03g}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -x -k "fused or grouped or transpose or tile_prune or one_pass or conv" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; echo pytest rc=$rc; tail -5 gpurun_out/${tag}_pytest.log
if [ "$rc" != 0 ]; then exit 1; fi
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
T="SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so"
step p1 $T SM_FUSED_WIDEP=1 $B
step p0 $T SM_FUSED_WIDEP=0 $B
step p1b $T SM_FUSED_WIDEP=1 $B
step p0b $T SM_FUSED_WIDEP=0 $B
step p1_f256 $T SM_FUSED_WIDEP=1 $B --fused-max-n 256
step p0_f256 $T SM_FUSED_WIDEP=0 $B --fused-max-n 256
# per-shape: wide family alone
for v in 1 0; do
  for shp in "784 256 2304" "784 256 1024" "3136 256 512" "196 512 4608" "784 512 1024"; do
    env $T SM_FUSED_WIDEP=$v timeout -k 10 100 python3 tools/time_fused.py $shp 32 2>/dev/null | sed "s/^/widep=$v /"
  done
done
