#!/bin/bash
# round 4, session am: rocprofv3 evidence for the fp32 split form: kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE), MfmaUtil
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04am_prof -- python3 tools/f32_split_profile.py 3 > gpurun_out/r04am_prof.log 2>&1; guard $? "rocprof stats"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04am_pmc_fetch -- python3 tools/f32_split_profile.py 2 > gpurun_out/r04am_pmc_fetch.log 2>&1; guard $? "pmc fetch"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r04am_pmc_write -- python3 tools/f32_split_profile.py 2 > gpurun_out/r04am_pmc_write.log 2>&1; guard $? "pmc write"
python3 tools/pmc_traffic.py gpurun_out/r04am_pmc_fetch gpurun_out/r04am_pmc_write gpurun_out/r04am_traffic.json
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r04am_pmc_mfma -- python3 tools/f32_split_profile.py 2 > gpurun_out/r04am_pmc_mfma.log 2>&1; guard $? "pmc mfma"
python3 tools/pmc_mfma.py gpurun_out/r04am_pmc_mfma gpurun_out/r04am_mfma.json
head -12 gpurun_out/r04am_prof/*/*kernel_stats.csv | cut -c1-170
python3 -c "
import json
t=json.load(open('gpurun_out/r04am_traffic.json'))
for k,v in t.items():
    if 'split' in k or 'gemm_f32' in k: print(k, v['launches_profiled'], 'MB/launch', round(v['hbm_bytes_per_launch']/1e6,1), 'us', round(v['avg_duration_us_in_pmc_pass'],1), 'GB/s', round(v['hbm_GBs']))
m=json.load(open('gpurun_out/r04am_mfma.json'))
for k,v in m.items():
    if 'split' in k or 'gemm_f32' in k: print(k, 'MfmaUtil %', round(v['mfma_util_percent'],1))"
