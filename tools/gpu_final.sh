#!/bin/bash
# final session of a round: the full gpu_round, then the other bench lines (fp32 config 2, config-4 sweep, bf16, 2-rank rehearsal)
set -o pipefail
tag=${1:-r03r}
bash tools/gpu_round.sh $tag || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/${tag}_bench_f32.json 2> gpurun_out/${tag}_bench_f32.err; echo "bench f32 rc=$?"
timeout -k 10 400 python bench.py --tables resnet50,resnet101,resnet152 --scaling lpt --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_cfg4.json 2> gpurun_out/${tag}_bench_cfg4.err; echo "bench cfg4 rc=$?"
timeout -k 10 400 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/${tag}_bench_bf16.json 2> gpurun_out/${tag}_bench_bf16.err; echo "bench bf16 rc=$?"
timeout -k 10 400 python bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 --no-extras > gpurun_out/${tag}_rehearse_gpus2.json 2> gpurun_out/${tag}_rehearse_gpus2.err; echo "rehearse rc=$?"
python3 -c "
import json
for n in ('bench','bench_f32','bench_cfg4','bench_bf16','rehearse_gpus2'):
    try:
        d=json.loads(open('gpurun_out/${tag}_'+n+'.json').read().strip().splitlines()[-1]); print(n, 'ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value']), 'n_gpus', d['n_gpus'], d['scaling'])
    except Exception as e: print(n, 'failed', e)
"
