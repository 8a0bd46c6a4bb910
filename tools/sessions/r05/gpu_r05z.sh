#!/bin/bash
# round 5, final session: tools/gpu_final.sh (full GPU suite, bench lines, rocprofv3 stats, PMC passes) + the per-shape table, the family counters,
# the emulated N = 2 / 4 / 8 runs
set -o pipefail
tag=${1:-r05z}
bash tools/gpu_final.sh $tag || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 400 python tools/sweep_grouped.py --table resnet50 --reps 3 > gpurun_out/${tag}_sweep_resnet50.txt 2> gpurun_out/${tag}_sweep_resnet50.err; guard $? sweep; tail -3 gpurun_out/${tag}_sweep_resnet50.txt
for n in 2 4 8; do
timeout -k 10 500 python bench.py --emulate-world $n > gpurun_out/${tag}_emu$n.json 2> gpurun_out/${tag}_emu$n.err; guard $? emu$n
done
python3 - <<PY
import json
out = {}
for n in (2, 4, 8):
    try:
        d = json.loads(open("gpurun_out/${tag}_emu%d.json" % n).read().strip().splitlines()[-1])
        out["N=%d" % n] = d
        print("emulated N =", n, "max_ms", round(d["max_ms"], 4), "spread", round(d["spread"], 3), "value", round(d["value"]))
    except Exception as e:
        print("emu", n, "failed", e)
json.dump(out, open("gpurun_out/${tag}_scale_emulated.json", "w"), indent=1)
PY
