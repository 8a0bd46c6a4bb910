#!/bin/bash
# round 5, session j: the thin form (depthwise layers): tests, the model-zoo bench lines and per-shape tables again
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "thin or model_zoo" > gpurun_out/r05j_pytest.log 2>&1; guard $? pytest; tail -4 gpurun_out/r05j_pytest.log
for t in mobilenetv2 mobilenetv3_small mobilenetv3_large; do
timeout -k 10 300 python bench.py --tables $t --no-cpu-baseline --no-extras > gpurun_out/r05j_bench_$t.json 2> gpurun_out/r05j_bench_$t.err; guard $? bench_$t
python3 -c "
import json
d=json.loads(open('gpurun_out/r05j_bench_$t.json').read().strip().splitlines()[-1]); print('$t: ms_per_step', round(d['ms_per_step'],4), 'GF/s', round(d['value']), 'verified', d.get('verified'))" || tail -3 gpurun_out/r05j_bench_$t.err
done
for t in mobilenetv2 mobilenetv3_large; do
timeout -k 10 400 python tools/sweep_grouped.py --table $t --reps 1 > gpurun_out/r05j_sweep_$t.txt 2> gpurun_out/r05j_sweep_$t.err; guard $? sweep_$t; grep -E "^ +[0-9]+ +1 " gpurun_out/r05j_sweep_$t.txt | head -8; tail -2 gpurun_out/r05j_sweep_$t.txt
done
