#!/bin/bash
# one bench.py run with extras (tools/gpu_bench_only.sh <tag> [bench args...])
set -o pipefail
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/${tag}_bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/${tag}_bench.json'))
print('ms_per_step',d['ms_per_step'],'value',d['value'])
s=d['stages']; print({k:(round(v,3) if isinstance(v,float) else v) for k,v in s.items() if k.endswith('_ms') or k.startswith('speedup') or k.startswith('pipelined')})
"
