#!/bin/bash
# round 5, session i: model-zoo parity test (fixed data), bench with the conv_step stage, measured plan costs (emulated N = 8), 2-rank rehearsal with the config-4 stage
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "model_zoo or streamk" > gpurun_out/r05i_pytest.log 2>&1; guard $? pytest; tail -4 gpurun_out/r05i_pytest.log
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/r05i_bench.json 2> gpurun_out/r05i_bench.err; guard $? bench; python3 -c "
import json
d=json.load(open('gpurun_out/r05i_bench.json'))
print('ms_per_step',d['ms_per_step']); print('conv_step', json.dumps(d['stages'].get('conv_step')))
print('yardstick', d['roofline']['yardstick'])
" || tail -5 gpurun_out/r05i_bench.err
timeout -k 10 500 python bench.py --emulate-world 8 > gpurun_out/r05i_emu8.json 2> gpurun_out/r05i_emu8.err; guard $? emu8; python3 -c "
import json
d=json.loads(open('gpurun_out/r05i_emu8.json').read().strip().splitlines()[-1]); print('emu8 max_ms', d['max_ms'], 'per_rank', [round(x,3) for x in d['per_rank_ms']], 'spread', round(d['spread'],3))" || tail -5 gpurun_out/r05i_emu8.err
timeout -k 10 600 python bench.py --gpus 2 --rehearse-gloo --steps 4 --warmup 1 --no-extras > gpurun_out/r05i_rehearse2.json 2> gpurun_out/r05i_rehearse2.err; guard $? rehearse2; python3 -c "
import json
d=json.loads(open('gpurun_out/r05i_rehearse2.json').read().strip().splitlines()[-1]); print('rehearse n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'plan_costs', d['config']['plan_costs']['source'], 'config4', json.dumps(d.get('stages',{}).get('config4_sweep'))[:400])" || tail -5 gpurun_out/r05i_rehearse2.err
