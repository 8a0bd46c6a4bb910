#!/bin/bash
# round 5, session l: the split-role big form at five tile heights against the dispatch rule (bit-identity checked by the tool)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so timeout -k 10 400 python tools/ab_big.py big2 3 > gpurun_out/r05l_ab_big2.txt 2>&1; guard $? ab; cat gpurun_out/r05l_ab_big2.txt
