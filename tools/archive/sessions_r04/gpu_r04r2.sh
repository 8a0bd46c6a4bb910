#!/bin/bash
# round 4, session r2: the dense twin (dense GEMM through the fused kernels' pipelines): parity of every GEMM test, per-shape A/B, bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "gemm or full_size or cpp or drivers or coo" > gpurun_out/r04r2_pytest.txt 2>&1; guard $? "pytest"; tail -8 gpurun_out/r04r2_pytest.txt
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so timeout -k 10 500 python tools/ab_dense.py resnet50 3 > gpurun_out/r04r2_ab_dense.txt 2> gpurun_out/r04r2_ab_dense.err; guard $? "ab dense"
cat gpurun_out/r04r2_ab_dense.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04r2_bench.json 2> gpurun_out/r04r2_bench.err; guard $? "bench"
python3 -c "
import json; d=json.loads(open('gpurun_out/r04r2_bench.json').read().strip().splitlines()[-1]); s=d['stages']
print('ms_per_step', round(d['ms_per_step'],4), 'verified', d.get('verified'))
for k in ('spmma_mul_ms','spmma_mul_grouped_ms','dense_gemm_rowmajor_ms','dense_gemm_rowmajor_grouped_ms','dense_gemm_batched_colmajor_ms','speedup_full_vs_dense_rowmajor','speedup_full_vs_dense_rowmajor_grouped','speedup_mul_vs_dense_rowmajor'): print(' ', k, s.get(k))"
