#!/bin/bash
# The PaiNN pass of scripts/run_pmc_mfma.sh alone (re-collected after the reverse message step moved to sender tiles).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_painn -o p -- python3 scripts/profile_painn.py force 50 > gpurun_out/pmc_mfma_painn.log 2>&1
echo "PaiNN pass done"
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
