#!/bin/bash
# cfconv build choice for launch groups: the automatic rule (rounds x cost) against the forced 4-wave (flag bit 3) and
# 8-wave (bit 2) builds, groups of 4 / 5 / 6 / 8 batches, four in flight
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-stream --no-config4-reference "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$LBL $*', round(d['value']/1e6), 'M edges/s', round(d['ms_per_step']*1e3,2), 'us/step  cfconv', round(r['avg_launch_us'],2), 'us frac', round(r['frac'],3), 'lone', round(d['single_forward_latency_ms']*1e3,1))"; }
for g in 5 6 8 4; do
  LBL="auto  " run --steps 200 --warmup 20 --group $g
  LBL="4-wave" MPENGINE_INFLIGHT_CFCONV_FLAGS=8 run --steps 200 --warmup 20 --group $g
  LBL="8-wave" MPENGINE_INFLIGHT_CFCONV_FLAGS=4 run --steps 200 --warmup 20 --group $g
done
LBL="auto  " run --steps 20 --warmup 5
LBL="auto  " run --steps 20 --warmup 5
