#!/bin/bash
# round 5, session g: does the stream-K form move the N = 8 prediction? (bench.py --emulate-world, hybrid plan, with / without)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for mode in off on; do
timeout -k 10 500 python bench.py --emulate-world 8 --streamk $mode > gpurun_out/r05g_emu8_$mode.json 2> gpurun_out/r05g_emu8_$mode.err; echo "emu8 $mode rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r05g_emu8_$mode.json').read().strip().splitlines()[-1]); print('$mode', 'speedup', d.get('speedup_vs_one_gpu'), 'max_ms', d['max_ms'], 'per_rank_ms', [round(x,3) for x in d['per_rank_ms']])"
done
