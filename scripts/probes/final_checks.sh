#!/bin/bash
# end-of-round checks on one box: every side bench still runs, the N = 2 rehearsals of bench.py (gloo, both workloads)
cd "$GRAFT_REPO_ROOT"
set -o pipefail
python scripts/bench_painn.py 64 --no-layers 2>/dev/null | tail -1 | cut -c150-420
python scripts/bench_gcn.py 2>/dev/null | tail -1 | cut -c1-300
python scripts/bench_moldyn.py 200 2>/dev/null | tail -1 | cut -c1-300
python scripts/bench_schnet_force.py 64 2>/dev/null | tail -1 | cut -c1-400
for w in config4 config2; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --workload $w 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2 gloo $w:', d['metric'], round(d['value']/1e6), 'M', d['scaling'], d['n_gpus'])"
done
