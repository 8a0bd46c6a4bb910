#!/bin/bash
# round 4, session ab: spmma<float> with f32_planes: one-pass prune + check (no blob) then the split multiply; header parity, fp32 bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "cpp or drivers" > gpurun_out/r04ab_pytest.txt 2>&1; guard $? "pytest"; tail -4 gpurun_out/r04ab_pytest.txt
timeout -k 10 400 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r04ab_bench_f32.json 2> gpurun_out/r04ab_bench_f32.err; guard $? "bench f32"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04ab_bench_f32.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],4), 'dense', round(d['stages']['dense_gemm_rowmajor_ms'],3), 'api exact', round(d['stages']['api_spmma_ms'],3)); print(json.dumps({k:v for k,v in d['stages'].get('f32_split').items() if k.endswith('_ms') or 'speedup' in k or 'layers' in k}, indent=1))"
