#!/bin/bash
# Round-3 kernel traces after the node-chain rework (rocprofv3 --kernel-trace --stats, condensed on the box into
# gpurun_out/stats/*.csv, copied to profiles/r03_*): config 2 lone forward, 4 in flight, launch groups, fresh-batch stream.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-config4-reference --no-stream --steps 300"
rocprofv3 --kernel-trace --stats -d gpurun_out/p3_c2_1 -o b -- $B --in-flight 1 > gpurun_out/p3_c2_1.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/p3_c2_4 -o b -- $B --group 1 --in-flight 4 > gpurun_out/p3_c2_4.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/p3_grp -o b -- $B > gpurun_out/p3_grp.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/p3_stream -o b -- python3 bench.py --workload stream > gpurun_out/p3_stream.log 2>&1
mkdir -p gpurun_out/stats
for d in p3_c2_1 p3_c2_4 p3_grp p3_stream; do
  db=$(find gpurun_out/$d -name "*_results.db" | head -1)
  [ -n "$db" ] && python3 scripts/rocprof_db_stats.py $db gpurun_out/stats/$d.csv "$d" > /dev/null && rm -rf gpurun_out/$d
done
ls -la gpurun_out/stats
