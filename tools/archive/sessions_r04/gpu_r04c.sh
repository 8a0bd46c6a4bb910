#!/bin/bash
# round 4, session c: stagger (big), priority (wide A loaders), PF=3, the rule-based dispatch inside the bench step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$PWD/sparsify.me_amd
SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 400 python - > gpurun_out/r04c_ab.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
sys.argv = ["ab_big.py", "wide,n128", "3"]
src = open("tools/ab_big.py").read()
src = src.replace('"wide": [("base", {}), ("big", {"SM_FUSED_BIG": "1"})],',
  '"wide": [("base", {}), ("aprio", {"SM_FUSED_APRIO": "1", "SM_FUSED_WIDEP": "0"}), ("pf3", {"SM_FUSED_WIDE_PF": "3", "SM_FUSED_WIDEP": "0"}), ("nowidep", {"SM_FUSED_WIDEP": "0"}), ("big", {"SM_FUSED_BIG": "1"}), ("big stag", {"SM_FUSED_BIG": "1", "SM_FUSED_BIG_STAG": "1"})],')
src = src.replace('"n128": [("base", {}), ("big nsb2", {"SM_FUSED_BIG": "2", "SM_FUSED_BIG_NSB": "2"}), ("big nsb3", {"SM_FUSED_BIG": "2", "SM_FUSED_BIG_NSB": "3"})],',
  '"n128": [("base", {}), ("big", {"SM_FUSED_BIG": "2"}), ("big stag", {"SM_FUSED_BIG": "2", "SM_FUSED_BIG_STAG": "1"})],')
exec(compile(src, "tools/ab_big.py", "exec"), {"__name__": "__main__", "__file__": os.path.abspath("tools/ab_big.py")})
PY
echo "ab rc=$?"; grep -v "bit-identical" gpurun_out/r04c_ab.txt | tail -12; grep "False" gpurun_out/r04c_ab.txt
for v in "" "SM_FUSED_BIG=8" "SM_FUSED_BIG=1" "SM_FUSED_BIG=8 SM_FUSED_BIG_STAG=1"; do
  tagv=$(echo "$v" | tr ' =' '__')
  env $v SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r04c_bench_${tagv}.json 2> gpurun_out/r04c_bench_${tagv}.err; echo "bench [$v] rc=$?"
  python3 -c "
import json; d=json.load(open('gpurun_out/r04c_bench_${tagv}.json')); print('  [$v] ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value']))"
done
for v in "" "SM_FUSED_BIG=8"; do
  env $v SPARSIFYME_LIB=$L/libsparsifyme_tuning.so timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r04c_bench2.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/r04c_bench2.json')); print('  again [$v] ms_per_step', round(d['ms_per_step'],4))"
done
