#!/bin/bash
# round 4, session al: check (valid operand) / STRIP prune with and without four loads in flight, same box, tuning library
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
for u in 1 0 1 0; do
  echo "== SM_PRUNE_UNROLL=$u"
  SM_PRUNE_UNROLL=$u SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so timeout -k 10 300 python tools/prune_rates.py 2> gpurun_out/r04al.err | grep -E "check|STRIP"; guard $? "rates $u"
done
