#!/bin/bash
# round 5, session f: stream-K tests on the tuned rule + bench with / without the stream-K workspaces
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests -m gpu -q --timeout 240 -x -k "streamk or grouped or bench_step" > gpurun_out/r05f_pytest.log 2>&1; guard $? pytest; tail -4 gpurun_out/r05f_pytest.log
for mode in on off on off; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --streamk $mode > gpurun_out/r05f_bench_$mode.json 2> gpurun_out/r05f_bench_$mode.err; guard $? bench_$mode
python3 -c "
import json
d=json.load(open('gpurun_out/r05f_bench_$mode.json')); print('streamk $mode: ms_per_step', round(d['ms_per_step'],4), 'verified', d.get('verified'), [v for v in d.get('verified_layers',[]) if v['family']=='sk'])"
done
timeout -k 10 400 python bench.py > gpurun_out/r05f_bench.json 2> gpurun_out/r05f_bench.err; guard $? bench; python3 -c "
import json
d=json.load(open('gpurun_out/r05f_bench.json'))
print('ms_per_step',d['ms_per_step'],'value',d['value'])
r=d['roofline']; print('roofline',r['kernel'],r['frac'],r['avg_launch_us'],'copy',r['yardstick']['device_copy_GBs'], 'traffic', r['traffic'], (r.get('traffic_source') or {}).get('stale'))
for k,v in r['families'].items(): print(' ',k,v['ms_per_step'],v['launches'],round(v['frac_of_hbm_peak'],3))
s=d['stages']; print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k.endswith('_ms') or k.startswith('speedup')})
"
