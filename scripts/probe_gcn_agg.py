"""Where does the GCN aggregate's time go at Cora size?  Variants of one launch: real graph vs degree-capped, weights
on/off, F = 64 / 128, plain segment-sum of pre-gathered rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.engine import _HipTimer
from gcnn_keras_amd.ragged import RaggedTensor

g = synth.cora_like_graph()
n, m = int(g["node_splits"][-1]), int(g["edge_splits"][-1])
timer = _HipTimer()


def run(idx_np, f, use_w, label):
    idx = RaggedTensor.from_numpy(idx_np, np.array([0, len(idx_np)], np.int64))
    nodes = RaggedTensor.from_numpy(np.zeros((n, 1), np.float32), g["node_splits"])
    plan = idx.index_plan(nodes)
    ptr, perm, _ = plan.csr(0)
    h = torch.randn(n, f, device="cuda")
    out = torch.empty(n, f, device="cuda")
    w = torch.rand(len(idx_np), device="cuda") if use_w else None
    send = plan.col(1).contiguous()
    ms = timer.time_ms(lambda: _ffi.call("mp_gather_segment_reduce_csr_f32", _ffi.MP_SUM, _ffi.ptr(h), n, f, _ffi.ptr(send),
                                         len(idx_np), _ffi.ptr(ptr), _ffi.ptr(perm), n, _ffi.ptr(w), 0, 1, 0.05,
                                         _ffi.ptr(out), _ffi.stream()), 50)
    deg = (ptr[1:] - ptr[:-1])
    print("%-46s F=%3d w=%d perm=%s  max_deg=%4d  %.2f us" % (label, f, use_w, perm is not None, int(deg.max()), ms * 1e3))


ei = g["edge_indices"]
run(ei, 64, True, "cora-like graph")
run(ei, 64, False, "cora-like graph")
run(ei, 128, True, "cora-like graph")
rng = np.random.default_rng(0)
reg = np.stack([np.repeat(np.arange(n), 5), rng.integers(0, n, 5 * n)], 1).astype(np.int64)   # every node 5 in-edges
run(reg, 64, True, "regular in-degree 5")
hub = reg.copy(); hub[:600, 0] = 0; hub = hub[np.lexsort((hub[:, 1], hub[:, 0]))]
run(hub, 64, True, "regular + one hub of ~600")
e = torch.empty(0, device="cuda")
ms = timer.time_ms(lambda: e.zero_(), 50)
print("empty torch launch %.2f us" % (ms * 1e3))
