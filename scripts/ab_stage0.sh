#!/bin/bash
# A/B of two builds of libmpengine.so (MPENGINE_LIB) on one box: rocprofv3 kernel averages of `bench.py --in-flight 1`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-config4-reference --steps 300 --in-flight 1"
for round in 1 2; do
  for tag in old new; do
    if [ $tag = old ]; then export MPENGINE_LIB=$GRAFT_REPO_ROOT/gcnn_keras_amd/csrc/libmpengine_old.so; else unset MPENGINE_LIB; fi
    rocprofv3 --kernel-trace --stats -d gpurun_out/ab_${tag}_$round -o b -- $B > gpurun_out/ab_${tag}_$round.log 2>&1
    db=$(find gpurun_out/ab_${tag}_$round -name "b_results.db" | head -1)
    echo "== $tag $round  $(tail -1 gpurun_out/ab_${tag}_$round.log | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print(round(l['single_forward_latency_ms']*1e3,2), 'us lone', round(l['ms_per_step']*1e3,2), 'us/step')")"
    python3 scripts/rocprof_db_stats.py $db gpurun_out/ab_${tag}_$round.csv x | grep "stage0\|painn_stage0" | cut -c1-90
    rm -rf gpurun_out/ab_${tag}_$round
  done
done
