#!/bin/bash
# Counter passes after the node-chain rework: the SQ matrix-pipe counters of config 2 (lone forward) and of a
# launch group's union, and the LDS counters of the same two runs (bank conflicts of the bf16-plane tiles).
# Condense with scripts/merge_pmc_mfma.py -> profiles/r03_pmc_mfma.json (SQ) and scripts/merge_pmc_lds.py (LDS).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 -L 2>&1 | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*" | sort -u > gpurun_out/pmc_counters_lds.txt || true
cat gpurun_out/pmc_counters_lds.txt
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
B="python3 bench.py --no-cpu-baseline --no-config4-reference --no-stream"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_c2 -o p -- $B --steps 200 --in-flight 1 > gpurun_out/pmc_mfma_c2.log 2>&1
echo "config 2 SQ pass done"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_grp -o p -- $B --steps 100 --in-flight 1 --group 5 > gpurun_out/pmc_mfma_grp.log 2>&1
echo "group SQ pass done"
LDS=$(grep -x "SQ_LDS_BANK_CONFLICT\|SQ_LDS_IDX_ACTIVE\|SQ_INSTS_LDS\|SQ_LDS_ADDR_CONFLICT\|SQ_LDS_UNALIGNED_STALL" gpurun_out/pmc_counters_lds.txt | tr '\n' ' ')
if [ -n "$LDS" ]; then
  rocprofv3 --pmc $LDS --kernel-trace --output-format csv -d gpurun_out/pmc_lds_grp -o p -- $B --steps 100 --in-flight 1 --group 5 > gpurun_out/pmc_lds_grp.log 2>&1
  echo "group LDS pass done ($LDS)"
fi
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
