#!/bin/bash
# the step under different stream counts / schedules on one box (no extras)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05aw}
for opt in "--streams 8" "--streams 4" "--streams 6" "--streams 12" "--streams 16" "--streams 8 --sched lpt" "--streams 8 --sched split" "--streams 8 --item-order small-first" "--streams 8"; do
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline $opt > gpurun_out/${tag}_tmp.json 2>/dev/null || exit 1
python3 -c "
import json
d=json.loads(open('gpurun_out/${tag}_tmp.json').read().strip().splitlines()[-1]); print('%-44s ms_per_step %.4f' % ('$opt', d['ms_per_step']))" | tee -a gpurun_out/${tag}_streams.txt
done
