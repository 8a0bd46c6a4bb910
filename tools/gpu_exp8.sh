#!/bin/bash
set -o pipefail
tag=${1:-r03n}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -x -k "coo or gemm" > gpurun_out/${tag}_pytest.log 2>&1; rc=$?; echo pytest rc=$rc; tail -5 gpurun_out/${tag}_pytest.log
if [ "$rc" != 0 ]; then grep -n "Error\|assert" gpurun_out/${tag}_pytest.log | head -20; exit 1; fi
timeout -k 10 300 python3 - <<'PY'
import sys, json
sys.path.insert(0, '.')
import torch, bench
import __graft_entry__ as ge
sm = ge.load_package()
dev = torch.device('cuda', 0)
r = bench.config5_stage(sm, torch, dev)
for s in r['shapes']:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items()})
c = bench.conv_path_stage(sm, torch, dev, 'f16')
print('conv table_weighted_ms', round(c['table_weighted_ms'], 3), [round(l['ms'] * 1e3, 1) for l in c['layers']])
PY
