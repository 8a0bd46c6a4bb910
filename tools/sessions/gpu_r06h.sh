#!/bin/bash
# round 6, session h: closed-loop balancing of the hybrid plan -- the real control flow with two gloo ranks on the one GPU, and the 8-rank emulation
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 --no-extras --rebalance 2 --detail gpurun_out/r06h_rehearse_detail.json > gpurun_out/r06h_rehearse.json 2> gpurun_out/r06h_rehearse.err; echo "rehearse rc=$?"; tail -3 gpurun_out/r06h_rehearse.err; python3 -c "
import json; d=json.load(open('gpurun_out/r06h_rehearse_detail.json')); print(d['n_gpus'], d['ms_per_step'], json.dumps(d['config']['plan_costs'].get('rebalance_rounds')))"
for n in 8 4; do
timeout -k 10 800 python bench.py --emulate-world $n --steps 10 --warmup 3 --rebalance 2 > gpurun_out/r06h_emu$n.json 2> gpurun_out/r06h_emu$n.err; echo "emu$n rc=$?"; tail -2 gpurun_out/r06h_emu$n.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r06h_emu$n.json').read().splitlines()[-1]); print('N=$n x', round(d['predicted_speedup_vs_n1'],3), 'max_ms', round(d['max_ms'],4), 'spread', round(d['spread'],3)); print(json.dumps(d['rebalance_rounds']))"
done
