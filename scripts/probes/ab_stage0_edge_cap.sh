cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-stream --no-config4-reference --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$LBL', round(d['value']/1e6), 'M edges/s', round(d['ms_per_step']*1e3,2), 'us/step lone', round(d['single_forward_latency_ms']*1e3,1))"; }
for rep in 1 2 3; do
  LBL="cap none" run
  LBL="cap 128 " MPENGINE_STAGE0_EDGE_CAP=128 run   # (the switch existed for this A/B only; 64 is built in now)
  LBL="cap 64  " MPENGINE_STAGE0_EDGE_CAP=64 run
done
