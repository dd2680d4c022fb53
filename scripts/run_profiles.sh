#!/bin/bash
# One profiling call on the GPU box: kernel traces (rocprofv3 --kernel-trace --stats) and separate --pmc passes for the
# workloads behind profiles/r02_*.  Outputs under gpurun_out/; condense with scripts/rocprof_db_stats.py and
# scripts/summarize_profiles.py r02.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-config4-reference --steps 300"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof5_c2_1 -o b -- $B --in-flight 1 > gpurun_out/prof5_c2_1.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof5_c2_4 -o b -- $B > gpurun_out/prof5_c2_4.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof5_sf -o sf -- python3 scripts/bench_schnet_force.py 64 --profile fork 200 > gpurun_out/prof5_sf.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof5_pn_ef -o painn -- python3 scripts/profile_painn.py force 200 > gpurun_out/prof5_pn_ef.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof5_pn_f -o painn -- python3 scripts/profile_painn.py forward 200 > gpurun_out/prof5_pn_f.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof5_gcn -o gcn -- python3 scripts/profile_gcn.py 300 > gpurun_out/prof5_gcn.log 2>&1
# condense on the box (the SQLite traces are 10-20 MB each; gpurun merges at most 64 MiB back) and drop the databases
mkdir -p gpurun_out/stats
for d in prof5_c2_1 prof5_c2_4 prof5_sf prof5_pn_ef prof5_pn_f prof5_gcn; do
  db=$(find gpurun_out/$d -name "*_results.db" | head -1)
  [ -n "$db" ] && python3 scripts/rocprof_db_stats.py $db gpurun_out/stats/$d.csv "$d" > /dev/null && rm -rf gpurun_out/$d
done
echo traces done
if [ -n "$SKIP_PMC" ]; then exit 0; fi
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc2_$n -o p -- python3 bench.py --no-cpu-baseline --no-config4-reference --steps 200 --in-flight 1 > gpurun_out/pmc2_$n.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc2_sf_$n -o p -- python3 scripts/bench_schnet_force.py 64 --profile fork 50 > gpurun_out/pmc2_sf_$n.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc2_pn_$n -o p -- python3 scripts/profile_painn.py force 50 > gpurun_out/pmc2_pn_$n.log 2>&1
  echo pmc $c done
done
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
