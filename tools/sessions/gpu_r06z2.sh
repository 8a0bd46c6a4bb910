#!/bin/bash
# round 6, final session (part 2): the fp32 line (config 2) with ITS counters, bf16, config 4, the two-rank rehearsal, the per-shape table,
# the API sequence per shape, the emulated N = 2 / 4 / 8 runs.   usage: bash tools/sessions/gpu_r06z2.sh <tag>
set -o pipefail
tag=${1:-r06z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" != 0 ]; then echo "$what failed; stopping"; exit 1; fi; }
timeout -k 10 400 python bench.py --dtype f32 --no-cpu-baseline --detail gpurun_out/${tag}_bench_f32_detail.json > gpurun_out/${tag}_bench_f32.json 2> gpurun_out/${tag}_bench_f32.err; guard $? "bench f32"
B="python3 bench.py --dtype f32 --eager --no-cpu-baseline --no-extras --detail gpurun_out/${tag}_scratch_detail.json"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_f32_fetch -- $B --steps 2 --warmup 1 > gpurun_out/${tag}_f32_fetch.log 2>&1; guard $? "f32 pmc fetch"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_f32_write -- $B --steps 2 --warmup 1 > gpurun_out/${tag}_f32_write.log 2>&1; guard $? "f32 pmc write"
python3 tools/pmc_traffic.py gpurun_out/${tag}_f32_fetch gpurun_out/${tag}_f32_write gpurun_out/${tag}_traffic_f32.json
rm -rf gpurun_out/${tag}_f32_fetch gpurun_out/${tag}_f32_write
timeout -k 10 400 python bench.py --tables resnet50,resnet101,resnet152 --scaling lpt --no-cpu-baseline --no-extras > gpurun_out/${tag}_bench_cfg4.json 2> gpurun_out/${tag}_bench_cfg4.err; guard $? "bench cfg4"
timeout -k 10 400 python bench.py --dtype bf16 --no-cpu-baseline --detail gpurun_out/${tag}_bench_bf16_detail.json > gpurun_out/${tag}_bench_bf16.json 2> gpurun_out/${tag}_bench_bf16.err; guard $? "bench bf16"
timeout -k 10 400 python bench.py --gpus 2 --rehearse-gloo --steps 3 --warmup 1 --no-extras > gpurun_out/${tag}_rehearse_gpus2.json 2> gpurun_out/${tag}_rehearse_gpus2.err; guard $? rehearse
timeout -k 10 400 python tools/sweep_grouped.py --table resnet50 --reps 3 > gpurun_out/${tag}_sweep_resnet50.txt 2> gpurun_out/${tag}_sweep_resnet50.err; guard $? sweep; tail -3 gpurun_out/${tag}_sweep_resnet50.txt
timeout -k 10 400 python tools/api_path_table.py > gpurun_out/${tag}_api_path.txt 2> gpurun_out/${tag}_api_path.err; guard $? "api table"; tail -1 gpurun_out/${tag}_api_path.txt
for n in 2 4 8; do
timeout -k 10 500 python bench.py --emulate-world $n --steps 10 --warmup 3 > gpurun_out/${tag}_emu$n.json 2> gpurun_out/${tag}_emu$n.err; guard $? emu$n
done
python3 - <<PY
import json
out = {}
for n in (2, 4, 8):
    d = json.loads(open("gpurun_out/${tag}_emu%d.json" % n).read().strip().splitlines()[-1])
    out["N=%d" % n] = d
    print("emulated N =", n, "max_ms", round(d["max_ms"], 4), "spread", round(d["spread"], 3), "x", round(d["predicted_speedup_vs_n1"], 3))
json.dump(out, open("gpurun_out/${tag}_scale_emulated.json", "w"), indent=1)
for n in ('bench_f32','bench_cfg4','bench_bf16','rehearse_gpus2'):
    d=json.loads(open('gpurun_out/${tag}_'+n+'.json').read().strip().splitlines()[-1]); print(n, 'ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value']), 'n_gpus', d['n_gpus'], d['scaling'])
PY
