#!/bin/bash
# round 4, session u: staged 2:4 matmul with 256 x 256 tiles (tuning A/B, grouped launches)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so timeout -k 10 500 python tools/ab_spmma.py 3 > gpurun_out/r04u_ab_spmma.txt 2> gpurun_out/r04u_ab_spmma.err; guard $? "ab spmma"
cat gpurun_out/r04u_ab_spmma.txt
