#!/bin/bash
# round 4, session n: big kernel with its DMA pieces spread over the sweep (A/B, tuning library); API-path table per shape
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so timeout -k 10 500 python tools/ab_big.py ilv 3 > gpurun_out/r04n_ab_ilv.txt 2>&1; guard $? "ab ilv"
cat gpurun_out/r04n_ab_ilv.txt | grep -v "bit-identical to .*: True"
timeout -k 10 400 python tools/api_path_table.py > gpurun_out/r04n_api_path.txt 2> gpurun_out/r04n_api_path.err; guard $? "api table"
cat gpurun_out/r04n_api_path.txt
