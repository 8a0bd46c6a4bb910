#!/bin/bash
# round 5, session e: stream-K with phase-aligned groups + lead cuts + deferred publish: tests, probe, A/B
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 400 python -m pytest tests -m gpu -q --timeout 240 -x -k "streamk" > gpurun_out/r05e_pytest.log 2>&1; guard $? pytest; tail -5 gpurun_out/r05e_pytest.log
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so timeout -k 10 300 python tools/sk_probe.py > gpurun_out/r05e_sk_probe.txt 2>&1; guard $? probe; cat gpurun_out/r05e_sk_probe.txt
timeout -k 10 240 python tools/ab_streamk.py 3 > gpurun_out/r05e_ab_rule.txt 2>&1; guard $? ab_rule; cat gpurun_out/r05e_ab_rule.txt
