#!/bin/bash
# A/B of cfconv builds for the in-flight bench: MPENGINE_INFLIGHT_CFCONV_FLAGS unset (4-wave) vs 4 (8 waves on one LDS image)
cd "$GRAFT_REPO_ROOT"
B="python bench.py --no-cpu-baseline --no-config4-reference"
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(l['value']/1e6,1), 'M edges/s', round(l['ms_per_step']*1e3,2), 'us/step  lone', round(l['single_forward_latency_ms']*1e3,2), 'us')"; }
for r in 1 2 3 4; do
  $B 2>/dev/null | show "flags0  "
  MPENGINE_INFLIGHT_CFCONV_FLAGS=4 $B 2>/dev/null | show "flags4  "
done
MPENGINE_INFLIGHT_CFCONV_FLAGS=4 $B --in-flight 3 2>/dev/null | show "flags4 x3"
MPENGINE_INFLIGHT_CFCONV_FLAGS=4 $B --in-flight 6 2>/dev/null | show "flags4 x6"
MPENGINE_INFLIGHT_CFCONV_FLAGS=4 $B --in-flight 8 2>/dev/null | show "flags4 x8"
