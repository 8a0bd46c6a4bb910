#!/bin/bash
# tune_direct.sh -- row tile of the direct fused kernel (128 vs 64 rows) on the n <= 128 layers (tuning library)
out=${1:-gpurun_out/tune}
mkdir -p $out
export SPARSIFYME_LIB=$PWD/sparsify.me_amd/libsparsifyme_tuning.so
cat > /tmp/direct_shapes.csv <<EOT
m,n,k,b
12544,64,64,32
12544,64,256,32
12544,64,576,32
12544,128,256,32
3136,128,512,32
3136,128,1152,32
EOT
cp /tmp/direct_shapes.csv tools/direct_shapes.csv
: > $out/tune_direct.txt
for bm in 128 64; do
  echo "== SM_FUSED_BM=$bm" >> $out/tune_direct.txt
  SM_FUSED_BM=$bm python tools/sweep.py --table ../tools/direct_shapes --only fused --reps 10 2>&1 | grep fused | grep -v "^fused" >> $out/tune_direct.txt
done
SM_FUSED_BM=64 timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "fused_equals_staged or full_size_properties_resnet50" > $out/pytest_bm64.log 2>&1
echo "bm64 pytest rc=$?" >> $out/tune_direct.txt; tail -2 $out/pytest_bm64.log >> $out/tune_direct.txt
