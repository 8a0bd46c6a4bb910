#!/bin/bash
# copy what a final session (tools/sessions/gpu_r06z.sh + gpu_r06z2.sh <tag>) left under gpurun_out/ into profiles/ under the names profiles/README.md
# lists, and make its counter files the *_latest.json that bench.py replays (they carry the hash of the library they were measured on)
tag=$1; [ -n "$tag" ] || { echo "usage: $0 <tag>"; exit 1; }
g=gpurun_out; p=profiles
c() { [ -f "$1" ] && cp "$1" "$2" || echo "missing $1"; }
c $g/${tag}_bench.json $p/bench_${tag}.json; c $g/${tag}_bench_detail.json $p/bench_${tag}_detail.json
for n in f32 bf16; do c $g/${tag}_bench_$n.json $p/bench_${n}_${tag}.json; c $g/${tag}_bench_${n}_detail.json $p/bench_${n}_${tag}_detail.json; done
c $g/${tag}_bench_cfg4.json $p/bench_cfg4_${tag}.json
c $g/${tag}_rehearse_gpus2.json $p/rehearse_gpus2_${tag}.json
c $g/${tag}_pytest_gpu.log $p/pytest_gpu_${tag}.txt
c $g/${tag}_traffic.json $p/traffic_${tag}.json; c $g/${tag}_traffic.json $p/traffic_latest.json
c $g/${tag}_traffic_f32.json $p/traffic_f32_${tag}.json; c $g/${tag}_traffic_f32.json $p/traffic_f32_latest.json
c $g/${tag}_mfma.json $p/mfma_util_${tag}.json; c $g/${tag}_mfma.json $p/mfma_util_latest.json
c $g/${tag}_prune_hbm.json $p/prune_hbm_${tag}.json
c $g/${tag}_sweep_resnet50.txt $p/sweep_${tag}_f16_resnet50.txt
c $g/${tag}_api_path.txt $p/api_path_${tag}.txt
c $g/${tag}_scale_emulated.json $p/scale_emulated_${tag}.json
c $g/${tag}_kernel_stats.csv $p/rocprof_${tag}_kernel_stats.csv
c $g/parity_margins.txt $p/parity_margins_${tag}.txt
ls $p | grep ${tag}
