#!/bin/bash
# round 4, session v: column-loop form of the fp32 split kernel (128 < n <= 256): parity, table with and without it
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "f32_split or cpp" > gpurun_out/r04v_pytest.txt 2>&1; guard $? "pytest"; tail -5 gpurun_out/r04v_pytest.txt
timeout -k 10 300 python tools/f32_split_table.py > gpurun_out/r04v_f32_split.txt 2> gpurun_out/r04v_f32_split.err; guard $? "table"
cat gpurun_out/r04v_f32_split.txt
SM_F32_SPLIT_COLS=0 SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so timeout -k 10 300 python tools/f32_split_table.py > gpurun_out/r04v_f32_split_nocols.txt 2> gpurun_out/r04v_f32_split_nocols.err; guard $? "table nocols"
grep -E " 256 | sums" gpurun_out/r04v_f32_split_nocols.txt
