#!/bin/bash
# Run-to-run spread of the default bench line on one box (ten runs)
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3 4 5 6 7 8 9 10; do
  python bench.py --no-cpu-baseline --no-config4-reference --steps ${1:-200} --warmup ${2:-20} 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(l['value']/1e6,1), 'M edges/s', round(l['ms_per_step']*1e3,2), 'us/step')"
done
