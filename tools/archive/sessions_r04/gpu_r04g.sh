#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "coo or cpp_drivers or cpp_headers" > gpurun_out/r04g_pytest.txt 2>&1; guard $? "pytest"; tail -5 gpurun_out/r04g_pytest.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err; guard $? bench; python3 -c "
import json
d=json.load(open('gpurun_out/r04g_bench.json'))
print('ms_per_step',d['ms_per_step'],'verified',d.get('verified'))
for r in d['stages']['config5_coo_spmm']['shapes']: print(' cfg5', r['m'],r['n'],r['k'], 'ms', round(r['ms'],4), 'frac', round(r['frac'],3), 'exact', round(r['ms_exact'],4), r['form'], r['range_flag'])
for r in d['stages']['bell_spmm']['shapes']: print(' bell', r['m'],r['n'],r['k'], round(r['ms'],4), round(r['frac'],3))
"
tail -2 gpurun_out/r04g_bench.err
for N in 8 4; do
  timeout -k 10 600 python bench.py --emulate-world $N --scaling hybrid --steps 10 --warmup 3 --settle-ms 100 > gpurun_out/r04g_emu_hybrid_$N.json 2> gpurun_out/r04g_emu_hybrid_$N.err; guard $? "emu hybrid $N"
  python3 -c "
import json; d=json.load(open('gpurun_out/r04g_emu_hybrid_$N.json')); print('hybrid N=$N', 'speedup', round(d['predicted_speedup_vs_n1'],3), 'max_ms', round(d['max_ms'],4), 'n1_ms', round(d['n1_ms'],4), 'spread', round(d['spread'],3), [round(x,3) for x in d['per_rank_ms']])"
done
