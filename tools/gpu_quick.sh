#!/bin/bash
# One short GPU-box session: selected GPU tests + the bench with and without grouped launches (tools/gpu_quick.sh <tag> [pytest -k expr])
set -o pipefail
tag=${1:-rXX}; kexpr="${2-grouped or fused_equals}"   # pass "" for the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -x -k "$kexpr" > gpurun_out/${tag}_pytest.log 2>&1; guard $? pytest; tail -3 gpurun_out/${tag}_pytest.log
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; guard $? bench; python3 -c "
import json,sys
d=json.load(open('gpurun_out/${tag}_bench.json'))
print('ms_per_step',d['ms_per_step'],'value',d['value'])
r=d['roofline']; print('roofline',r['kernel'],r['frac'],r['avg_launch_us'],'copy',r['yardstick']['device_copy_GBs'])
for k,v in r['families'].items(): print(' ',k,v['ms_per_step'],v['launches'],round(v['frac_of_hbm_peak'],3))
s=d['stages']; print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k.endswith('_ms') or k.startswith('speedup')})
"
timeout -k 10 300 python bench.py --group off --no-cpu-baseline > gpurun_out/${tag}_bench_nogroup.json 2> gpurun_out/${tag}_bench_nogroup.err; guard $? bench_nogroup; python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_bench_nogroup.json'))
print('nogroup ms_per_step',d['ms_per_step'])
r=d['roofline']
for k,v in r['families'].items(): print(' ',k,v['ms_per_step'],v['launches'],round(v['frac_of_hbm_peak'],3))
"
