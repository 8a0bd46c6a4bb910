#!/bin/bash
# every few-tile shape of the ResNet-50 step under each kernel family that can take it, same box (tuning library; grouped as the step launches them)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05ba}
export SPARSIFYME_LIB=sparsify.me_amd/libsparsifyme_tuning.so
S="3136,512,128,4 784,1024,256,6 3136,256,512,1 784,256,2304,6 784,256,1024,5 784,512,1024,1 196,512,4608,3 196,2048,512,3 196,512,2048,2"
for r in 1 2; do
echo "== default rule"; timeout -k 10 250 python tools/ab_env.py SM_NONE 0 $S 2>&1 | grep -v amdgpu || exit 1
echo "== wide forced (SM_FUSED_WIDE=1)"; SM_FUSED_WIDE=1 timeout -k 10 250 python tools/ab_env.py SM_NONE 0 $S 2>&1 | grep -v amdgpu || exit 1
echo "== big forced (SM_FUSED_BIG=1)"; SM_FUSED_BIG=1 timeout -k 10 250 python tools/ab_env.py SM_NONE 0 $S 2>&1 | grep -v amdgpu || exit 1
echo "== no big, astat where it applies (SM_FUSED_BIG=0)"; SM_FUSED_BIG=0 timeout -k 10 250 python tools/ab_env.py SM_NONE 0 $S 2>&1 | grep -v amdgpu || exit 1
done | tee gpurun_out/${tag}_families.txt
