#!/bin/bash
# tune_rega.sh -- register-A form of the direct fused kernel (SM_FUSED_REGA=<PF>, tuning library) against the default
out=${1:-gpurun_out/tune}
mkdir -p $out
export SPARSIFYME_LIB=$PWD/sparsify.me_amd/libsparsifyme_tuning.so
: > $out/tune_rega.txt
SM_FUSED_REGA=2 timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "fused_equals_staged or full_size_properties_resnet50" > $out/pytest_rega.log 2>&1
rc=$?; echo "rega pytest rc=$rc" >> $out/tune_rega.txt; tail -2 $out/pytest_rega.log >> $out/tune_rega.txt
[ $rc = 0 ] || exit 1
for pf in 0 2 3; do
  echo "== SM_FUSED_REGA=$pf" >> $out/tune_rega.txt
  SM_FUSED_REGA=$pf timeout -k 10 200 python tools/sweep.py --table tools/direct_shapes.csv --only fused --reps 10 2>&1 | grep fused | grep -v "^fused" >> $out/tune_rega.txt || exit 1
done
