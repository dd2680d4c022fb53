#!/bin/bash
# headline sweep: batches per launch group x groups in flight (200 steps), final state of round 3
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stream --no-config4-reference "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', round(d['value']/1e6), 'M edges/s', round(d['ms_per_step']*1e3,2), 'us/step')"; }
for cfg in "5 4" "5 5" "5 6" "5 3" "4 5" "4 6" "5 4"; do set -- $cfg; run --group $1 --in-flight $2; done
