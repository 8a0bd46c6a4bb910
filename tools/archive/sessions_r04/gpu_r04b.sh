#!/bin/bash
# round 4, session b: stamps of big / wide / direct, B-fragment prefetch depth, RCCL world-1 test, emulate-world smoke
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$PWD/sparsify.me_amd
SPARSIFYME_LIB=$L/libsparsifyme_stamp.so timeout -k 10 300 python tools/stamp_shapes.py \
  "784,256,2304,6:SM_FUSED_BIG=1" "784,256,2304,6:SM_FUSED_BIG=1;SM_FUSED_BIG_PF=2" "784,256,2304,6:SM_FUSED_WIDEP=0" \
  "196,512,2048,2:SM_FUSED_BIG=1" "196,512,2048,2:SM_FUSED_WIDEP=0" \
  "3136,128,1152,4:SM_FUSED_BIG=2" "3136,128,1152,4" "12544,64,576,3" > gpurun_out/r04b_stamp.txt 2>&1; echo "stamp rc=$?"
grep STAMP gpurun_out/r04b_stamp.txt
SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 300 python - > gpurun_out/r04b_pf.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
sys.argv = ["ab_big.py", "wide", "3"]
import runpy
src = open("tools/ab_big.py").read().replace('("big", {"SM_FUSED_BIG": "1"})]', '("big", {"SM_FUSED_BIG": "1"}), ("big pf2", {"SM_FUSED_BIG": "1", "SM_FUSED_BIG_PF": "2"})]', 1)
exec(compile(src, "tools/ab_big.py", "exec"), {"__name__": "__main__", "__file__": os.path.abspath("tools/ab_big.py")})
PY
echo "pf rc=$?"; grep -v "bit-identical" gpurun_out/r04b_pf.txt | tail -8; grep -c "True" gpurun_out/r04b_pf.txt; grep "False" gpurun_out/r04b_pf.txt
timeout -k 10 300 python -m pytest tests/test_bench_multirank.py -m gpu -q -k "rccl" > gpurun_out/r04b_rccl.txt 2>&1; echo "rccl rc=$?"; tail -3 gpurun_out/r04b_rccl.txt
timeout -k 10 400 python bench.py --emulate-world 2 --steps 10 --warmup 3 --settle-ms 100 > gpurun_out/r04b_emu2.json 2> gpurun_out/r04b_emu2.err; echo "emu rc=$?"; cat gpurun_out/r04b_emu2.json | cut -c1-900
