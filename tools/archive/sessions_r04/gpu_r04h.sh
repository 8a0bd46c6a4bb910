#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$PWD/sparsify.me_amd
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 400 python - > gpurun_out/r04h_ab.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
sys.argv = ["ab_big.py", "wide,n128,astat", "3"]
src = open("tools/ab_big.py").read()
src = src.replace('"wide": [("base", {}), ("big", {"SM_FUSED_BIG": "1"})],',
  '"wide": [("base", {"SM_FUSED_BIG": "0"}), ("big", {"SM_FUSED_BIG": "1"}), ("big pp", {"SM_FUSED_BIG": "1", "SM_FUSED_BIG_PP": "1"})],')
src = src.replace('"n128": [("base", {}), ("big nsb2", {"SM_FUSED_BIG": "2", "SM_FUSED_BIG_NSB": "2"}), ("big nsb3", {"SM_FUSED_BIG": "2", "SM_FUSED_BIG_NSB": "3"})],',
  '"n128": [("base", {"SM_FUSED_BIG": "0"}), ("big", {"SM_FUSED_BIG": "2"}), ("big pp", {"SM_FUSED_BIG": "2", "SM_FUSED_BIG_PP": "1"})],')
src = src.replace('"astat": [("base", {}), ("big", {"SM_FUSED_BIG": "1", "SM_FUSED_ASTAT": "0"})],',
  '"astat": [("base", {"SM_FUSED_BIG": "0"}), ("big", {"SM_FUSED_BIG": "1", "SM_FUSED_ASTAT": "0"}), ("big pp", {"SM_FUSED_BIG": "1", "SM_FUSED_ASTAT": "0", "SM_FUSED_BIG_PP": "1"})],')
exec(compile(src, "tools/ab_big.py", "exec"), {"__name__": "__main__", "__file__": os.path.abspath("tools/ab_big.py")})
PY
guard $? ab; grep -v "bit-identical" gpurun_out/r04h_ab.txt | tail -14; grep "False" gpurun_out/r04h_ab.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "coo_fast" > gpurun_out/r04h_pytest.txt 2>&1; guard $? pytest; tail -3 gpurun_out/r04h_pytest.txt
timeout -k 10 300 python - > gpurun_out/r04h_coo.txt 2>&1 <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
import __graft_entry__ as ge
sm = ge.load_package()
r = bench.config5_stage(sm, torch, torch.device("cuda", 0))
for x in r["shapes"]: print(x["m"], x["n"], x["k"], "ms", round(x["ms"], 4), "frac", round(x["frac"], 3), "exact", round(x["ms_exact"], 4), x["form"], x["range_flag"])
PY
guard $? coo; cat gpurun_out/r04h_coo.txt | tail -18
