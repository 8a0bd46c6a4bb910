#!/bin/bash
# round 6, final session (part 1 of 2; part 2 = gpu_r06z2.sh, a second gpurun call: the limit of one call is 20 minutes):
# full GPU suite, the f16 bench (contract line + detail), rocprofv3 kernel stats, PMC traffic (FETCH / WRITE in separate passes) and MFMA-busy
# passes of the same step, the prune-step counters.   usage: bash tools/sessions/gpu_r06z.sh <tag>
set -o pipefail
tag=${1:-r06z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" != 0 ]; then echo "$what failed; stopping"; exit 1; fi; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${tag}_pytest_gpu.log 2>&1; guard $? pytest; tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout -k 10 400 python bench.py --detail gpurun_out/${tag}_bench_detail.json > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; guard $? bench; cat gpurun_out/${tag}_bench.json
B="python3 bench.py --eager --no-cpu-baseline --no-extras --detail gpurun_out/${tag}_scratch_detail.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- $B --streams 1 --steps 5 --warmup 2 > gpurun_out/${tag}_prof.log 2>&1; guard $? "rocprof stats"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- $B --steps 2 --warmup 1 > gpurun_out/${tag}_pmc_fetch.log 2>&1; guard $? "pmc fetch"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- $B --steps 2 --warmup 1 > gpurun_out/${tag}_pmc_write.log 2>&1; guard $? "pmc write"
python3 tools/pmc_traffic.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_traffic.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_mfma -- $B --streams 1 --steps 2 --warmup 1 > gpurun_out/${tag}_pmc_mfma.log 2>&1; guard $? "pmc mfma"
python3 tools/pmc_mfma.py gpurun_out/${tag}_pmc_mfma gpurun_out/${tag}_mfma.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_prune_fetch -- python3 tools/prune_profile.py 2 > gpurun_out/${tag}_prune_fetch.log 2>&1; guard $? "prune pmc fetch"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_prune_write -- python3 tools/prune_profile.py 2 > gpurun_out/${tag}_prune_write.log 2>&1; guard $? "prune pmc write"
python3 tools/pmc_traffic.py gpurun_out/${tag}_prune_fetch gpurun_out/${tag}_prune_write gpurun_out/${tag}_prune_hbm.json
# keep what travels back small: the raw pass directories stay on the box except the kernel-stats csv
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_mfma gpurun_out/${tag}_prune_fetch gpurun_out/${tag}_prune_write
ls gpurun_out | grep ${tag}
