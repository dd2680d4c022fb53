#!/bin/bash
# A/B of two library builds on one box: libmpengine.so (activation tiles split once by the producer) against
# libmpengine_old.so (every consuming wave splits the tile): parity tests on the new build, then the bench alternating.
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_fused.py tests/test_gpu_schnet.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2 || exit 1
B="python bench.py --no-cpu-baseline --no-config4-reference --no-stream"
show() { python -c "
import json,sys
l=json.loads([x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')][-1])
print('$1', round(l['value']/1e6,1), 'M edges/s', round(l['ms_per_step']*1e3,2), 'us/step  lone', round(l['single_forward_latency_ms']*1e3,2), 'us')"; }
OLD=$PWD/gcnn_keras_amd/csrc/libmpengine_old.so
for r in 1 2 3; do
  $B 2>/dev/null | show "new        "
  MPENGINE_LIB=$OLD $B 2>/dev/null | show "old        "
done
$B --workload config4 2>/dev/null | show "new config4"
MPENGINE_LIB=$OLD $B --workload config4 2>/dev/null | show "old config4"
