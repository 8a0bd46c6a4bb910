#!/bin/bash
# round 4, session q: grouped launches of the staged 2:4 matmul (parity + bench stage); header parity with the fp32 split option; drivers
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "spmma_grouped or test_spmma or cpp or drivers or full_size" > gpurun_out/r04q_pytest.txt 2>&1; guard $? "pytest"; tail -8 gpurun_out/r04q_pytest.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04q_bench.json 2> gpurun_out/r04q_bench.err; guard $? "bench"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04q_bench.json').read().strip().splitlines()[-1]); s=d['stages']
print('ms_per_step', round(d['ms_per_step'],4), 'verified', d.get('verified'))
for k in ('spmma_mul_ms','spmma_mul_grouped_ms','dense_gemm_rowmajor_ms','dense_gemm_rowmajor_grouped_ms','dense_gemm_batched_colmajor_ms','speedup_mul_vs_dense_rowmajor','speedup_mul_grouped_vs_dense_rowmajor_grouped','speedup_mul_grouped_vs_dense_batched','api_spmma_ms','api_spmma_one_kernel_ms'): print(' ', k, s.get(k))"
