#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in 0 6 7 8; do
SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so SM_COOFAST_TILE=$t timeout -k 10 200 python3 - <<'PY'
import sys, os
sys.path.insert(0, '.')
import torch, bench
import __graft_entry__ as ge
sm = ge.load_package()
r = bench.config5_stage(sm, torch, torch.device('cuda', 0))
print('tile', os.environ['SM_COOFAST_TILE'], [round(s['ms_fast_form'] * 1e3, 1) for s in r['shapes']])
PY
done
