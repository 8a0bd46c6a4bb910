#!/bin/bash
# round 4, session d: single-GPU emulation of the N-GPU plans (bench.py --emulate-world) + the new parity tests
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "bench_step_launches or full_size_properties_resnet50 or coo_config5 or coo_fast_config5" > gpurun_out/r04d_pytest.txt 2>&1; guard $? pytest; tail -3 gpurun_out/r04d_pytest.txt
for N in 2 4 8; do
  timeout -k 10 600 python bench.py --emulate-world $N --scaling hybrid --steps 10 --warmup 3 --settle-ms 100 > gpurun_out/r04d_emu_hybrid_$N.json 2> gpurun_out/r04d_emu_hybrid_$N.err; guard $? "emu hybrid $N"
  python3 -c "
import json; d=json.load(open('gpurun_out/r04d_emu_hybrid_$N.json')); print('hybrid N=$N', 'speedup', round(d['predicted_speedup_vs_n1'],3), 'max_ms', round(d['max_ms'],4), 'n1_ms', round(d['n1_ms'],4), 'spread', round(d['spread'],3), [round(x,3) for x in d['per_rank_ms']])"
done
for N in 2 8; do
  timeout -k 10 600 python bench.py --emulate-world $N --scaling strong --steps 10 --warmup 3 --settle-ms 100 > gpurun_out/r04d_emu_strong_$N.json 2> gpurun_out/r04d_emu_strong_$N.err; guard $? "emu strong $N"
  python3 -c "
import json; d=json.load(open('gpurun_out/r04d_emu_strong_$N.json')); print('strong N=$N', 'speedup', round(d['predicted_speedup_vs_n1'],3), 'max_ms', round(d['max_ms'],4), 'n1_ms', round(d['n1_ms'],4))"
done
timeout -k 10 900 python bench.py --emulate-world 8 --scaling lpt --tables resnet50,resnet101,resnet152 --steps 5 --warmup 2 --settle-ms 100 > gpurun_out/r04d_emu_lpt_cfg4_8.json 2> gpurun_out/r04d_emu_lpt_cfg4_8.err; guard $? "emu lpt cfg4 8"
python3 -c "
import json; d=json.load(open('gpurun_out/r04d_emu_lpt_cfg4_8.json')); print('lpt cfg4 N=8', 'speedup', round(d['predicted_speedup_vs_n1'],3), 'max_ms', round(d['max_ms'],4), 'n1_ms', round(d['n1_ms'],4), 'spread', round(d['spread'],3), [round(x,3) for x in d['per_rank_ms']])"
