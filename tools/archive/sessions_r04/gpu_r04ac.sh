#!/bin/bash
# round 4, session ac: the convolution layers by three routes (implicit GEMM, im2col + fused, im2col-to-blob + staged matmul)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 400 python tools/conv_routes.py > gpurun_out/r04ac_conv_routes.txt 2> gpurun_out/r04ac_conv_routes.err; guard $? "conv routes"
cat gpurun_out/r04ac_conv_routes.txt; tail -3 gpurun_out/r04ac_conv_routes.err
