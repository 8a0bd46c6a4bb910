#!/bin/bash
# round 4, session m: counter passes per kernel family on the final library, per-shape table launched as the step launches,
# and one bench line with the round's traffic / MfmaUtil tables in place
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
bash tools/pmc_families.sh r04 > gpurun_out/r04m_pmc.log 2>&1; guard $? "pmc_families"
timeout -k 10 500 python tools/sweep_grouped.py --table resnet50 > gpurun_out/r04m_sweep_resnet50.txt 2> gpurun_out/r04m_sweep.err; guard $? "sweep"
tail -3 gpurun_out/r04m_sweep_resnet50.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04m_bench.json 2> gpurun_out/r04m_bench.err; guard $? "bench"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04m_bench.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],4), 'roofline', d['roofline'])"
cat gpurun_out/r04_pmc_families.txt
