#!/bin/bash
# round 6, session i: the hybrid plan with GROUP-aware costs (a rank's g instances of a shape cost what their grouped launch measures): emulated 2 / 4 / 8
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for n in 8 4 2; do
timeout -k 10 600 python bench.py --emulate-world $n --steps 10 --warmup 3 > gpurun_out/r06i_emu$n.json 2> gpurun_out/r06i_emu$n.err; echo "emu$n rc=$?"; tail -2 gpurun_out/r06i_emu$n.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r06i_emu$n.json').read().splitlines()[-1]); print('N=$n x', round(d['predicted_speedup_vs_n1'],3), 'n1', round(d['n1_ms'],4), 'max_ms', round(d['max_ms'],4), 'spread', round(d['spread'],3), [round(x,4) for x in d['per_rank_ms']])"
done
timeout -k 10 300 python bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 --no-extras > gpurun_out/r06i_rehearse.json 2> gpurun_out/r06i_rehearse.err; echo "rehearse rc=$?"; tail -2 gpurun_out/r06i_rehearse.err; cut -c1-300 gpurun_out/r06i_rehearse.json
