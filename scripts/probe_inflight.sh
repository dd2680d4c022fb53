#!/bin/bash
# edges/s of the default bench line against the number of batches in flight (two step counts)
cd "$GRAFT_REPO_ROOT"
for n in 2 3 4 5 6 8; do
  for k in "200 20" "20 5"; do
    set -- $k
    python bench.py --no-cpu-baseline --no-config4-reference --in-flight $n --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('in flight $n  steps $1 ', round(l['value']/1e6,1), 'M edges/s', round(l['ms_per_step']*1e3,2), 'us/step')"
  done
done
