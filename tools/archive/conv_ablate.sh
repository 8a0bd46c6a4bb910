#!/bin/bash
# conv_ablate.sh -- what bounds the implicit-GEMM kernel: the probe with parts of the kernel switched off (tuning library)
out=${1:-gpurun_out/tune}
mkdir -p $out
export SPARSIFYME_LIB=$PWD/sparsify.me_amd/libsparsifyme_tuning.so CONV_PROBE_ONLY=implicit
: > $out/conv_ablate.txt
for ab in 0 1 2 4 8 3 5 7 15; do
  echo "== SM_CONV_ABLATE=$ab (1 no patch DMA, 2 no gather, 4 no B DMA, 8 no SMFMAC)" >> $out/conv_ablate.txt
  SM_CONV_ABLATE=$ab python tools/conv_probe.py 2>&1 | grep -v amdgpu.ids | grep "|" >> $out/conv_ablate.txt
done
