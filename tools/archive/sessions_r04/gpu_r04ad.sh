#!/bin/bash
# round 4, session ad: routed convolution entry point: parity, bench stage
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "conv" > gpurun_out/r04ad_pytest.txt 2>&1; guard $? "pytest"; tail -4 gpurun_out/r04ad_pytest.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04ad_bench.json 2> gpurun_out/r04ad_bench.err; guard $? "bench"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04ad_bench.json').read().strip().splitlines()[-1]); c=d['stages']['conv_path']
print('ms_per_step', round(d['ms_per_step'],4), 'conv table-weighted', round(c['table_weighted_ms'],4), 'routed', round(c['table_weighted_routed_ms'],4))
for r in c['layers']: print('  ', r['m'], r['n'], r['k'], 'x', r['count'], round(r['ms']*1e3,1), '->', round(r['ms_routed']*1e3,1), r['route'])"
