#!/bin/bash
# A/B of the node-kernel builds for the in-flight bench: bf16-piece images (default, 368 registers: one workgroup per CU)
# vs FP32 matrix instructions (MPENGINE_NODE_BF16=0: 250 registers for the MID chain, two workgroups per CU)
cd "$GRAFT_REPO_ROOT"
B="python bench.py --no-cpu-baseline --no-config4-reference"
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(l['value']/1e6,1), 'M edges/s', round(l['ms_per_step']*1e3,2), 'us/step  lone', round(l['single_forward_latency_ms']*1e3,2), 'us')"; }
for r in 1 2 3; do
  $B 2>/dev/null | show "bf16 nodes        "
  MPENGINE_NODE_BF16=0 $B 2>/dev/null | show "fp32 nodes        "
  MPENGINE_NODE_BF16=0 MPENGINE_INFLIGHT_CFCONV_FLAGS=4 $B 2>/dev/null | show "fp32 nodes+8wave "
done
