#!/bin/bash
# tune_nw.sh -- waves per workgroup of the direct fused kernel (4 x 32 rows vs 8 x 16 rows; tuning library)
out=${1:-gpurun_out/tune}
mkdir -p $out
export SPARSIFYME_LIB=$PWD/sparsify.me_amd/libsparsifyme_tuning.so
: > $out/tune_nw.txt
SM_FUSED_NW=8 timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "fused_equals_staged or full_size_properties_resnet50" > $out/pytest_nw8.log 2>&1
rc=$?; echo "nw8 pytest rc=$rc" >> $out/tune_nw.txt; tail -2 $out/pytest_nw8.log >> $out/tune_nw.txt
[ $rc = 0 ] || exit 1
for nw in 4 8; do
  echo "== SM_FUSED_NW=$nw" >> $out/tune_nw.txt
  SM_FUSED_NW=$nw timeout -k 10 200 python tools/sweep.py --table tools/direct_shapes.csv --only fused --reps 10 2>&1 | grep fused | grep -v "^fused" >> $out/tune_nw.txt || exit 1
done
