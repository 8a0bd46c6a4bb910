#!/bin/bash
# round 4, session ae: why do two processes on one GPU finish the table faster than one?  hardware queues x streams x batch split in one
# process, and the two-process rehearsal with 20 steps
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
run() { q=$1; shift
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/r04ae_b.json 2> gpurun_out/r04ae_b.err; guard $? "bench q=$q $*"
  python3 -c "
import json; d=json.loads(open('gpurun_out/r04ae_b.json').read().strip().splitlines()[-1]); print('  [queues $q $*] ms_per_step', round(d['ms_per_step'],4), 'n_gpus', d['n_gpus'])"; }
run 4 --streams 8
run 8 --streams 8
run 8 --streams 16
run 8 --streams 16 --batch-split 2
run 6 --streams 12 --batch-split 2
run 4 --gpus 2 --rehearse-gloo --steps 20 --warmup 5
run 4 --gpus 2 --rehearse-gloo --steps 20 --warmup 5
run 4 --streams 8
