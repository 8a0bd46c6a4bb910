#!/bin/bash
# round 4, session o: fp32 2:4 product on the sparse matrix instruction (bf16 splits): parity + the fp32 bench line;
# one-kernel prune + multiply at n <= 256: parity + the API-path table
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "f32_split or prune_spmma or spmma_fused_f32" > gpurun_out/r04o_pytest.txt 2>&1; guard $? "pytest"; tail -15 gpurun_out/r04o_pytest.txt
timeout -k 10 400 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r04o_bench_f32.json 2> gpurun_out/r04o_bench_f32.err; guard $? "bench f32"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04o_bench_f32.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],4), 'dense', round(d['stages']['dense_gemm_rowmajor_ms'],3)); print(json.dumps(d['stages'].get('f32_split'), indent=1))"
timeout -k 10 400 python tools/api_path_table.py > gpurun_out/r04o_api_path.txt 2> gpurun_out/r04o_api_path.err; guard $? "api table"
cat gpurun_out/r04o_api_path.txt
