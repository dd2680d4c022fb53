#!/bin/bash
# node chains of a union launch on half the CUs (flag bit 9, set by engine.SchnetForward for launches in flight) against all
# CUs (MPENGINE_INFLIGHT_NODE_HALF=0): default bench, 200 and 20 steps, three alternations
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_fused.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2 || exit 1
run() { python bench.py --no-cpu-baseline --no-stream --no-config4-reference "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$LBL $*', round(d['value']/1e6), 'M edges/s', round(d['ms_per_step']*1e3,2), 'us/step lone', round(d['single_forward_latency_ms']*1e3,1))"; }
for rep in 1 2 3; do
  LBL="half CUs" run --steps 200 --warmup 20
  LBL="all CUs " MPENGINE_INFLIGHT_NODE_HALF=0 run --steps 200 --warmup 20
done
for rep in 1 2; do
  LBL="half CUs" run --steps 20 --warmup 5
  LBL="all CUs " MPENGINE_INFLIGHT_NODE_HALF=0 run --steps 20 --warmup 5
done
