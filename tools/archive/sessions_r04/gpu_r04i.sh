#!/bin/bash
# round 4, session i: fp32 fused 2:4 register-form variants (config 2), COO fast form after the launch trimming
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$PWD/sparsify.me_amd
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
for mode in 0 1 2 3 4; do
  SM_F32_FUSED_MODE=$mode SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "spmma_fused_f32 or resnet18_f32" > gpurun_out/r04i_pytest_f32_$mode.txt 2>&1; guard $? "pytest f32 mode $mode"; tail -1 gpurun_out/r04i_pytest_f32_$mode.txt
  SM_F32_FUSED_MODE=$mode SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/r04i_bench_f32_$mode.json 2> gpurun_out/r04i_bench_f32_$mode.err; guard $? "bench f32 mode $mode"
  python3 -c "
import json; d=json.load(open('gpurun_out/r04i_bench_f32_$mode.json')); s=d['stages']
print('  f32 mode $mode ms_per_step', round(d['ms_per_step'],4), 'dense_rm', round(s['dense_gemm_rowmajor_ms'],4), 'ratio', round(s['speedup_full_vs_dense_rowmajor'],3), 'roofline', round(d['roofline']['frac'],3))"
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "coo_fast" > gpurun_out/r04i_pytest_coo.txt 2>&1; guard $? pytest; tail -2 gpurun_out/r04i_pytest_coo.txt
timeout -k 10 300 python - > gpurun_out/r04i_coo.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
import __graft_entry__ as ge
sm = ge.load_package()
r = bench.config5_stage(sm, torch, torch.device("cuda", 0))
for x in r["shapes"]: print(x["m"], x["n"], x["k"], "ms", round(x["ms"], 4), "frac", round(x["frac"], 3), "exact", round(x["ms_exact"], 4), x["form"], x["range_flag"])
PY
guard $? coo; tail -17 gpurun_out/r04i_coo.txt
