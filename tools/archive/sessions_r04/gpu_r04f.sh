#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "prune_spmma or bench_step_launches" > gpurun_out/r04f_pytest_new.txt 2>&1; guard $? "pytest new"; tail -3 gpurun_out/r04f_pytest_new.txt
timeout -k 10 400 python bench.py > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err; guard $? bench; python3 -c "
import json,sys
d=json.load(open('gpurun_out/r04f_bench.json'))
print('ms_per_step',d['ms_per_step'],'value',d['value'],'verified',d.get('verified'))
r=d['roofline']; print('roofline',r['kernel'],r['frac'],r['avg_launch_us'],'copy',r['yardstick']['device_copy_GBs'])
for k,v in r['families'].items(): print(' ',k,round(v['ms_per_step'],4),v['launches'],v['layers'],round(v['frac_of_hbm_peak'],3))
s=d['stages']; print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k.endswith('_ms') or k.startswith('speedup')})
"
tail -3 gpurun_out/r04f_bench.err
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/r04f_pytest_gpu.txt 2>&1; guard $? "pytest all"; tail -4 gpurun_out/r04f_pytest_gpu.txt
