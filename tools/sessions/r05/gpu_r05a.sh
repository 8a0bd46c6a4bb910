#!/bin/bash
# round 5, session a: the new tests (stream-K, fp32 column-loop last stage, conv 4-byte fallback / workspace query) + stream-K A/B
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 500 python -m pytest tests -m gpu -q --timeout 240 -x -k "streamk or cols_last_stage or conv_spmma or f32_split" > gpurun_out/r05a_pytest.log 2>&1; guard $? pytest; tail -5 gpurun_out/r05a_pytest.log
timeout -k 10 240 python tools/ab_streamk.py 3 > gpurun_out/r05a_ab_rule.txt 2>&1; guard $? ab_rule; cat gpurun_out/r05a_ab_rule.txt
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so SM_FUSED_SK=2 timeout -k 10 240 python tools/ab_streamk.py 3 > gpurun_out/r05a_ab_all.txt 2>&1; guard $? ab_all; cat gpurun_out/r05a_ab_all.txt
