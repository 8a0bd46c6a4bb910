#!/bin/bash
set -o pipefail
tag=${1:-r03h}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
step() { label=$1; shift
  out=$(env "$@" 2>gpurun_out/${tag}_${label}.err); rc=$?
  echo "$out" > gpurun_out/${tag}_${label}.json
  python3 -c "
import json
try:
    d=json.loads(open('gpurun_out/${tag}_${label}.json').read().strip().splitlines()[-1]); print('$label', 'ms_per_step', round(d['ms_per_step'],4))
except Exception as e: print('$label', 'failed', e)
"
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$label hit its limit; stopping"; exit 1; fi
}
B="timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline"
step base $B
step split2 $B --sched split --big-streams 2
step split3 $B --sched split --big-streams 3
step split4 $B --sched split --big-streams 4
step split1 $B --sched split --big-streams 1
step base_b $B
step split2_b $B --sched split --big-streams 2
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo bench rc=$?; python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print('ms_per_step',d['ms_per_step'],'value',d['value'])
r=d['roofline']; print('roofline',r['kernel'],r['frac'],r['avg_launch_us'],'copy',r['yardstick']['device_copy_GBs'])
for k,v in r['families'].items(): print(' ',k,v['ms_per_step'],v['launches'],v['layers'],round(v['frac_of_hbm_peak'],3))
s=d['stages']; print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k.endswith('_ms') or k.startswith('speedup')})
"
