#!/bin/bash
# One GPU-box session: full GPU test-suite, bench, rocprof kernel stats, PMC traffic passes.
# usage: bash tools/gpu_round.sh <tag>     (outputs under gpurun_out/<tag>_*)
set -o pipefail
tag=${1:-rXX}; extra="${2:-}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# a step that was killed at its limit ends the session: no further GPU step is started after it
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${tag}_pytest_gpu.log 2>&1; guard $? "pytest"; tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; guard $? "bench"; cat gpurun_out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --eager --streams 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${tag}_prof.log 2>&1; guard $? "rocprof stats"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_pmc_fetch.log 2>&1; guard $? "pmc fetch"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_pmc_write.log 2>&1; guard $? "pmc write"
python3 tools/pmc_traffic.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_traffic.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc_mfma -- python3 bench.py --eager --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_pmc_mfma.log 2>&1; guard $? "pmc mfma"
python3 tools/pmc_mfma.py gpurun_out/${tag}_pmc_mfma gpurun_out/${tag}_mfma.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_prune_fetch -- python3 tools/prune_profile.py 2 > gpurun_out/${tag}_prune_fetch.log 2>&1; guard $? "prune pmc fetch"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_prune_write -- python3 tools/prune_profile.py 2 > gpurun_out/${tag}_prune_write.log 2>&1; guard $? "prune pmc write"
python3 tools/pmc_traffic.py gpurun_out/${tag}_prune_fetch gpurun_out/${tag}_prune_write gpurun_out/${tag}_prune_hbm.json
