#!/bin/bash
# Counter passes over one representative launch per kernel family (tools/family_shapes.py): which limit each family
# hits.  usage: bash tools/pmc_families.sh <tag>  -> gpurun_out/<tag>_pmc_<pass>/ + gpurun_out/<tag>_pmc_summary.txt
set -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
guard() { rc=$1; what=$2; echo "$what rc=$rc"; if [ "$rc" = 124 ] || [ "$rc" = 137 ]; then echo "$what hit its limit; stopping"; exit 1; fi; }
run() { name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${tag}_pmc_${name} -- python3 tools/family_shapes.py 3 > gpurun_out/${tag}_pmc_${name}.log 2>&1; guard $? "pmc $name"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
run sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES
# (a six-counter TCP pass aborts rocprofv3 and every TCC_* pass tried hung until its limit on this pool: two small TCP passes only)
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
for p in sq1 sq2 sq3 tcp1 tcp2; do echo "== pass $p"; python3 tools/pmc_generic.py gpurun_out/${tag}_pmc_${p}; done > gpurun_out/${tag}_pmc_summary.txt 2>&1
python3 tools/pmc_family_table.py gpurun_out/${tag}_pmc_sq1 gpurun_out/${tag}_pmc_sq2 gpurun_out/${tag}_pmc_sq3 gpurun_out/${tag}_pmc_tcp1 gpurun_out/${tag}_pmc_tcp2 > gpurun_out/${tag}_pmc_families.txt 2>&1
# kernel durations of the same launches (from the first pass's kernel trace)
python3 - <<PY >> gpurun_out/${tag}_pmc_summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/${tag}_pmc_sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== kernel durations under the sq1 pass (us, average)")
for n, v in acc.items():
    print(n, round(sum(v) / len(v), 1), "launches", len(v))
PY
tail -40 gpurun_out/${tag}_pmc_summary.txt
