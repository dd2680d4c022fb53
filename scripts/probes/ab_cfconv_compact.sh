cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --no-stream --no-config4-reference --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$LBL', round(d['value']/1e6), 'M edges/s', round(d['ms_per_step']*1e3,2), 'us/step lone', round(d['single_forward_latency_ms']*1e3,1))"; }
for rep in 1 2 3; do
  LBL="default       " run
  LBL="cfconv compact" MPENGINE_INFLIGHT_CFCONV_FLAGS=16 run
done
