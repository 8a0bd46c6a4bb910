#!/bin/bash
# round 6, session c: whole GPU suite (TILE winner search regrouped, COO exact loads unpredicated, f32 split default, alias, ADVICE fixes),
# the f16 and f32 benches, and the 8-rank emulation with the refined hybrid plan + per-share costs
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06c_pytest.log 2>&1; rc=$?; tail -8 gpurun_out/r06c_pytest.log; [ $rc -ne 0 ] && exit $rc
bash tools/gpu_bench_only.sh r06c || exit 1
timeout -k 10 400 python bench.py --dtype f32 --detail gpurun_out/r06c_bench_f32_detail.json > gpurun_out/r06c_bench_f32.json 2> gpurun_out/r06c_bench_f32.err; echo "bench f32 rc=$?"; cat gpurun_out/r06c_bench_f32.json
timeout -k 10 500 python bench.py --emulate-world 8 --steps 10 --warmup 3 > gpurun_out/r06c_emu8.json 2> gpurun_out/r06c_emu8.err; echo "emu8 rc=$?"; cat gpurun_out/r06c_emu8.json
