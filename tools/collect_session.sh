#!/bin/bash
# copy what a final session (tools/sessions/gpu_r05z.sh <tag>) left under gpurun_out/ into profiles/ under the names profiles/README.md lists,
# and make its counter files the *_latest.json that bench.py replays (they carry the hash of the library they were measured on)
tag=$1; [ -n "$tag" ] || { echo "usage: $0 <tag>"; exit 1; }
g=gpurun_out; p=profiles
cp $g/${tag}_bench.json $p/bench_${tag}.json
for n in f32 cfg4 bf16; do cp $g/${tag}_bench_$n.json $p/bench_${n}_${tag}.json; done
cp $g/${tag}_rehearse_gpus2.json $p/rehearse_gpus2_${tag}.json
cp $g/${tag}_pytest_gpu.log $p/pytest_gpu_${tag}.txt
cp $g/${tag}_traffic.json $p/traffic_${tag}.json; cp $g/${tag}_traffic.json $p/traffic_latest.json
cp $g/${tag}_mfma.json $p/mfma_util_${tag}.json; cp $g/${tag}_mfma.json $p/mfma_util_latest.json
cp $g/${tag}_prune_hbm.json $p/prune_hbm_${tag}.json
cp $g/${tag}_sweep_resnet50.txt $p/sweep_${tag}_f16_resnet50.txt
cp $g/${tag}_scale_emulated.json $p/scale_emulated_${tag}.json
f=$(find $g/${tag}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $p/rocprof_${tag}_kernel_stats.csv
[ -f $g/parity_margins.txt ] && cp $g/parity_margins.txt $p/parity_margins_${tag}.txt
ls $p | grep ${tag}
