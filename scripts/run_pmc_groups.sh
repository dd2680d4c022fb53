#!/bin/bash
# Counter passes of the DEFAULT bench run (config 2 served in launch groups: five 128-graph batches per launch sequence,
# four groups in flight): HBM traffic (FETCH_SIZE, WRITE_SIZE - separate passes) and the SQ matrix-pipe counters.  Condense
# with scripts/merge_pmc_groups.py -> profiles/r03_pmc_hbm_traffic.json, profiles/r03_pmc_mfma.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-config4-reference --no-stream --steps 100 --in-flight 1 --group 5"
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_grp_$n -o p -- $B > gpurun_out/pmc_grp_$n.log 2>&1
  echo "pmc $c done"
done
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_grp -o p -- $B > gpurun_out/pmc_mfma_grp.log 2>&1
echo "sq pass done"
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
