#!/bin/bash
# tune_coo_j.sh -- vectors per workgroup of the packed COO kernel (tuning library): per-kernel us from rocprofv3
out=${1:-gpurun_out/tune}
mkdir -p $out
export SPARSIFYME_LIB=$PWD/sparsify.me_amd/libsparsifyme_tuning.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out/tune_coo_j.txt
for j in 0 8 16 32; do
  echo "== SM_SPMM_PK_J=$j" >> $out/tune_coo_j.txt
  rm -rf $out/prof_j$j
  SM_SPMM_PK_J=$j timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_j$j -- python3 tools/coo_probe.py 2 > $out/prof_j$j.log 2>&1 || exit 1
  grep -h "packed_k\|csr_lds" $out/prof_j$j/*/*kernel_stats.csv | cut -c1-48,180-250 >> $out/tune_coo_j.txt
done
