#!/bin/bash
# Separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the fused GCN forward and of the config-2 SchNet forward; merge with
# scripts/merge_pmc.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc3_gcn_$n -o p -- python3 scripts/profile_gcn.py 50 > gpurun_out/pmc3_gcn_$n.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc2_$n -o p -- python3 bench.py --no-cpu-baseline --no-config4-reference --steps 200 --in-flight 1 > gpurun_out/pmc2_$n.log 2>&1
  echo pmc $c done
done
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
