#!/bin/bash
# round 4, session aa: step composition -- every layer's batch as S independent problems (more, smaller work items over the streams)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
for v in "--batch-split 1" "--batch-split 2" "--batch-split 4" "--batch-split 2 --streams 12" "--batch-split 2 --streams 16" "--batch-split 1" "--batch-split 2"; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline $v > gpurun_out/r04aa_b.json 2> gpurun_out/r04aa_b.err; guard $? "bench $v"
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04aa_b.json').read().strip().splitlines()[-1]); print('  [$v] ms_per_step', round(d['ms_per_step'],4), 'verified', d.get('verified'))"
done
