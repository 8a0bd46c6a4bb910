#!/bin/bash
# round 6, session d: the residency experiment (tools/residency_probe.py on the tuning library) and the API sequence per shape
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so timeout -k 10 400 python tools/residency_probe.py > gpurun_out/r06d_residency.txt 2> gpurun_out/r06d_residency.err; echo "residency rc=$?"; cat gpurun_out/r06d_residency.txt; tail -3 gpurun_out/r06d_residency.err
timeout -k 10 400 python tools/api_path_table.py > gpurun_out/r06d_api_path.txt 2> gpurun_out/r06d_api_path.err; echo "api table rc=$?"; cat gpurun_out/r06d_api_path.txt; tail -3 gpurun_out/r06d_api_path.err
