cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stats
rocprofv3 --kernel-trace --stats -d gpurun_out/p3_pn_ef -o painn -- python3 scripts/profile_painn.py force 200 > gpurun_out/p3_pn_ef.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/p3_pn_f -o painn -- python3 scripts/profile_painn.py forward 200 > gpurun_out/p3_pn_f.log 2>&1
for d in p3_pn_ef p3_pn_f; do
  db=$(find gpurun_out/$d -name "*_results.db" | head -1)
  [ -n "$db" ] && python3 scripts/rocprof_db_stats.py $db gpurun_out/stats/$d.csv "$d" > /dev/null && rm -rf gpurun_out/$d
done
head -20 gpurun_out/stats/p3_pn_ef.csv
python scripts/bench_painn.py --no-layers > gpurun_out/bp.log 2>&1; tail -1 gpurun_out/bp.log
