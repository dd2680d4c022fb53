#!/bin/bash
# Runtime environment knobs against the headline bench (4 batches in flight and one at a time): one JSON summary line each.
cd "$GRAFT_REPO_ROOT"
B="python bench.py --no-cpu-baseline --no-config4-reference"
show() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(l['value']/1e6,1), 'M edges/s', round(l['ms_per_step']*1e3,2), 'us/step  lone', round(l['single_forward_latency_ms']*1e3,2), 'us')"; }
$B 2>/dev/null | show base
$B 2>/dev/null | show base_again
HIP_FORCE_DEV_KERNARG=1 $B 2>/dev/null | show dev_kernarg1
HIP_FORCE_DEV_KERNARG=0 $B 2>/dev/null | show dev_kernarg0
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $B 2>/dev/null | show pkt_capture1
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B 2>/dev/null | show pkt_capture0
GPU_MAX_HW_QUEUES=16 $B 2>/dev/null | show hwq16
GPU_MAX_HW_QUEUES=4 $B 2>/dev/null | show hwq4
HSA_ENABLE_INTERRUPT=0 $B 2>/dev/null | show no_interrupt
