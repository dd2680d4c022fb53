python scripts/probes/profile_fresh_group.py > gpurun_out/prof_fresh4.log 2>&1; grep "edges/s\|arena\|torch.empty\|torch.zeros" gpurun_out/prof_fresh4.log
python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -2
for cfg in "5 4" "10 2" "4 5" "20 1" "7 3" "5 4"; do set -- $cfg; for rep in 1 2; do python bench.py --steps 20 --warmup 5 --group $1 --in-flight $2 --no-cpu-baseline --no-stream --no-config4-reference 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('group $1 inflight $2:', round(d['value']/1e6), 'M edges/s', d['ms_per_step'])"; done; done
