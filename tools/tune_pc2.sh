#!/bin/bash
# tune_pc2.sh -- the cross-stage-prefetch producer/consumer 2:4 matmul (spmma_f16_pc2_kernel) against the default
# dispatch on the few-tile shapes, and its parity tests (tuning library).
out=${1:-gpurun_out/tune}
mkdir -p $out
export SPARSIFYME_LIB=$PWD/sparsify.me_amd/libsparsifyme_tuning.so
: > $out/tune_pc2.txt
for cfg in "" 128x4 128x8 256x4 256x8; do
  echo "== SM_SPMMA_PC2=$cfg" >> $out/tune_pc2.txt
  if [ -z "$cfg" ]; then unset SM_SPMMA_PC2; else export SM_SPMMA_PC2=$cfg; fi
  python tools/sweep.py --table ../tools/tune_shapes --only spmma --reps 10 2>&1 | grep spmma | grep -v "^spmma" >> $out/tune_pc2.txt
done
for cfg in 128x4 256x8; do
  export SM_SPMMA_PC2=$cfg
  timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "spmma_f16_vs_oracle or lane_maps or k_tail or alpha_beta or spmma_bf16 or full_size_properties_resnet50" > $out/pytest_pc2_$cfg.log 2>&1
  echo "pc2 $cfg pytest rc=$?" >> $out/tune_pc2.txt
  tail -3 $out/pytest_pc2_$cfg.log >> $out/tune_pc2.txt
done
