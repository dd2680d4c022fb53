#!/bin/bash
# Kernel trace of ONE launch group at a time (bench.py --group 5 --in-flight 1): the durations of the union launches
# without other groups on the GPU, listed per (kernel, grid size) - the cfconv line with the large grid is the launch the
# bench line's roofline object times with HIP events.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-config4-reference --no-stream --steps 300"
rocprofv3 --kernel-trace --stats -d gpurun_out/p3_grp1 -o b -- $B --group 5 --in-flight 1 > gpurun_out/p3_grp1.log 2>&1
mkdir -p gpurun_out/stats
db=$(find gpurun_out/p3_grp1 -name "*_results.db" | head -1)
python3 scripts/rocprof_db_stats.py $db gpurun_out/stats/p3_grp1.csv "p3_grp1" --by-grid > /dev/null && rm -rf gpurun_out/p3_grp1
cut -c1-150 gpurun_out/stats/p3_grp1.csv | head -24
grep '^{' gpurun_out/p3_grp1.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench roofline avg_launch_us', r['avg_launch_us'], 'frac', r['frac'], 'value', d['value'])"
