#!/bin/bash
# one bench.py run with extras (tools/gpu_bench_only.sh <tag> [bench args...]): stdout = the compact contract line
# (gpurun_out/<tag>_bench.json), the detail object = gpurun_out/<tag>_bench_detail.json
set -o pipefail
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --detail gpurun_out/${tag}_bench_detail.json "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${tag}_bench.err
python3 -c "
import json
raw=open('gpurun_out/${tag}_bench.json').read()
print('stdout lines', len(raw.splitlines()), 'bytes', len(raw))
d=json.loads(raw.splitlines()[-1])
print('ms_per_step',d['ms_per_step'],'value',d['value'], 'roofline', {k:d['roofline'][k] for k in ('kernel','frac','avg_launch_us','traffic')})
print(d['stages'])
"
