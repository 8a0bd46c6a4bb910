#!/bin/bash
# round 6, session g: which unit is busy in the few-tile kernels?  One counter pass over a representative launch per family (tools/family_shapes.py):
# LDS array cycles and bank conflicts, LDS / MFMA / VALU instruction counts and busy cycles, against the CU's busy cycles.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/r06g_pmc_lds -- python3 tools/family_shapes.py 3 > gpurun_out/r06g_pmc_lds.log 2>&1; echo "pass lds rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r06g_pmc_valu -- python3 tools/family_shapes.py 3 > gpurun_out/r06g_pmc_valu.log 2>&1; echo "pass valu rc=$?"
python3 tools/pmc_generic.py gpurun_out/r06g_pmc_lds > gpurun_out/r06g_units.txt 2>&1
echo "== second pass" >> gpurun_out/r06g_units.txt
python3 tools/pmc_generic.py gpurun_out/r06g_pmc_valu >> gpurun_out/r06g_units.txt 2>&1
rm -rf gpurun_out/r06g_pmc_lds gpurun_out/r06g_pmc_valu
cat gpurun_out/r06g_units.txt
