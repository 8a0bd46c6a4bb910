#!/bin/bash
set -o pipefail
tag=${1:-r03f}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
step() { label=$1; shift
  out=$(env "$@" 2>gpurun_out/${tag}_${label}.err); rc=$?
  echo "$out" > gpurun_out/${tag}_${label}.json
  python3 -c "
import json
try:
    d=json.loads(open('gpurun_out/${tag}_${label}.json').read().strip().splitlines()[-1]); print('$label', 'ms_per_step', round(d['ms_per_step'],4))
except Exception as e: print('$label', 'failed', e)
"
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$label hit its limit; stopping"; exit 1; fi
}
B="timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --fused-max-n 512"
step g1_s8 $B --streams 8
step gps_s4 $B --streams 4 --graphs per-stream
step gps_s8 $B --streams 8 --graphs per-stream
step gps_s12 $B --streams 12 --graphs per-stream
step eager_s8 $B --streams 8 --eager
step eager_s4 $B --streams 4 --eager
step g1_s8_b $B --streams 8
step gps_s8_b $B --streams 8 --graphs per-stream
# timeline of one step, graph replay vs per-stream graphs
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_kt_single -- python3 bench.py --no-extras --no-cpu-baseline --fused-max-n 512 --streams 8 --steps 3 --warmup 1 --settle-ms 0 > gpurun_out/${tag}_kt_single.log 2>&1; echo kt_single rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_kt_ps -- python3 bench.py --no-extras --no-cpu-baseline --fused-max-n 512 --streams 8 --graphs per-stream --steps 3 --warmup 1 --settle-ms 0 > gpurun_out/${tag}_kt_ps.log 2>&1; echo kt_ps rc=$?
for v in single ps; do echo "== $v"; python3 tools/ktrace_step.py gpurun_out/${tag}_kt_$v 24; done > gpurun_out/${tag}_ktrace.txt 2>&1; cat gpurun_out/${tag}_ktrace.txt
# TCP counters one pass at a time, short limit (the 6-counter pass aborted rocprofv3)
for c in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_pmc_$n -- python3 tools/family_shapes.py 2 > gpurun_out/${tag}_pmc_$n.log 2>&1; rc=$?; echo "pmc $n rc=$rc"
  if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "pmc $n hit its limit; stopping"; exit 1; fi
  python3 tools/pmc_generic.py gpurun_out/${tag}_pmc_$n 2>/dev/null | grep -v fill_uniform | cut -c1-400
done
