#!/bin/bash
# tune_pc.sh -- ring depth / loader waves / consumer waves of the producer-consumer 2:4 matmul on the few-tile shapes
# (tuning library: make -C sparsify.me_amd tuning).  Output: gpurun_out/<dir>/tune_pc.txt
out=${1:-gpurun_out/tune}
mkdir -p $out
export SPARSIFYME_LIB=$PWD/sparsify.me_amd/libsparsifyme_tuning.so
: > $out/tune_pc.txt
for cfg in "" 4x3 4x4 4x5 4x6 8x3 8x4 8x5 8x6 4x3x8 4x4x8 4x5x8 8x3x8 8x4x8 8x5x8 256x3 256x4; do
  echo "== SM_SPMMA_PC=$cfg" >> $out/tune_pc.txt
  if [ -z "$cfg" ]; then unset SM_SPMMA_PC; else export SM_SPMMA_PC=$cfg; fi
  python tools/sweep.py --table ../tools/tune_shapes --only spmma --reps 10 2>&1 | grep spmma | grep -v "^spmma" >> $out/tune_pc.txt
done
