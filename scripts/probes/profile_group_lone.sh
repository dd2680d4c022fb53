set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-config4-reference --no-stream --steps 300"
rocprofv3 --kernel-trace --stats -d gpurun_out/p3_grp1 -o b -- $B --group 5 --in-flight 1 > gpurun_out/p3_grp1.log 2>&1
mkdir -p gpurun_out/stats
db=$(find gpurun_out/p3_grp1 -name "*_results.db" | head -1)
python3 scripts/rocprof_db_stats.py $db gpurun_out/stats/p3_grp1.csv "p3_grp1" > /dev/null && rm -rf gpurun_out/p3_grp1
cat gpurun_out/stats/p3_grp1.csv | cut -c1-150 | head -14
