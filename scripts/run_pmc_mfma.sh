#!/bin/bash
# Counter evidence for the matrix-pipe claims: one rocprofv3 --pmc pass per workload with the SQ
# counters of the MFMA pipe (the program directly after `--`, no trace domains besides the kernel trace - see the
# profiling rules of the pool).  Condense with scripts/merge_pmc_mfma.py -> profiles/r03_pmc_mfma.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/pmc_counters_available.txt 2>&1 || true
grep -o -i "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU\b\|SQ_WAIT_INST_LDS\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_VALU\|GRBM_GUI_ACTIVE" \
  gpurun_out/pmc_counters_available.txt | sort -u > gpurun_out/pmc_counters_mfma.txt || true
cat gpurun_out/pmc_counters_mfma.txt
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
B="python3 bench.py --no-cpu-baseline --no-config4-reference"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_c2 -o p -- $B --steps 200 --in-flight 1 > gpurun_out/pmc_mfma_c2.log 2>&1
echo "config 2 pass done"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_shard -o p -- $B --workload config4 --total-graphs 12500 --steps 10 --warmup 2 > gpurun_out/pmc_mfma_shard.log 2>&1
echo "12500-graph shard pass done"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_painn -o p -- python3 scripts/profile_painn.py force 50 > gpurun_out/pmc_mfma_painn.log 2>&1
echo "PaiNN pass done"
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
python3 scripts/merge_pmc_mfma.py || true
du -sh gpurun_out
