#!/bin/bash
# round 4, session aj: single-GPU emulation of the default N-GPU plan (hybrid) on the final library, N = 2, 4, 8
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
for N in 2 4 8; do
  timeout -k 10 600 python bench.py --emulate-world $N --scaling hybrid --steps 10 --warmup 3 --settle-ms 100 > gpurun_out/r04aj_emu_hybrid_$N.json 2> gpurun_out/r04aj_emu_hybrid_$N.err; guard $? "emu hybrid $N"
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04aj_emu_hybrid_$N.json').read().strip().splitlines()[-1]); print('  N=$N predicted speedup', round(d['predicted_speedup_vs_n1'],3), 'max_ms', round(d['max_ms'],4), 'n1_ms', round(d['n1_ms'],4), 'spread', round(d['spread'],3))"
done
