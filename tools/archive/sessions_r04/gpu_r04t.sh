#!/bin/bash
# round 4, session t: dense fp32 GEMM by bfloat16 pieces: parity, per-shape table, fp32 bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "f32_split" > gpurun_out/r04t_pytest.txt 2>&1; guard $? "pytest"; tail -5 gpurun_out/r04t_pytest.txt
timeout -k 10 300 python tools/f32_split_table.py > gpurun_out/r04t_f32_split.txt 2> gpurun_out/r04t_f32_split.err; guard $? "table"
cat gpurun_out/r04t_f32_split.txt
timeout -k 10 400 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r04t_bench_f32.json 2> gpurun_out/r04t_bench_f32.err; guard $? "bench f32"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04t_bench_f32.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],4), 'dense', round(d['stages']['dense_gemm_rowmajor_ms'],3)); print(json.dumps({k:v for k,v in d['stages'].get('f32_split').items() if k!='kernel'}, indent=1))"
