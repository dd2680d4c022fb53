"""BASELINE config 5 (GCN node classification on one Cora-shaped graph: 2708 nodes, 13264 directed edges incl. self loops,
F = 1433 -> 64, depth 3, head [64, 32, 7] softmax) on one MI355X.

    python scripts/bench_gcn.py

Prints ONE JSON line: latency of ``GCN.make_model(...)(inputs)`` - the fused route (csrc/mp_gcn.hip: 1 + depth tile
launches replayed from a HIP graph) and the layer sequence (``fused=False``, one engine call per Keras layer) - and the
kernels that carry the data, timed alone with HIP events, each against the roofline that bounds it: the fused route's
input launch (reads the (2708,1433) feature matrix: 15.5 MB) and layer launch (aggregate + Dense), and the layer path's
first Dense (split-K) and gather x weight -> segment-sum -> ReLU aggregate (125 B/edge).  A parity configuration, not
the headline line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.engine import _HipTimer
from gcnn_keras_amd.literature import GCN
from gcnn_keras_amd.ragged import RaggedTensor

HBM_PEAK = 8000.0


def timeit(fn, n, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    g = synth.cora_like_graph()
    n, m, f = int(g["node_splits"][-1]), int(g["edge_splits"][-1]), int(g["node_attributes"].shape[1])
    ins = [RaggedTensor.from_numpy(g["node_attributes"], g["node_splits"]),
           RaggedTensor.from_numpy(g["edge_weights"], g["edge_splits"]),
           RaggedTensor.from_numpy(g["edge_indices"], g["edge_splits"])]
    model = GCN.make_model(inputs=[{"shape": (None, f), "name": "node_attributes", "dtype": "float32", "ragged": True},
                                   {"shape": (None, 1), "name": "edge_weights", "dtype": "float32", "ragged": True},
                                   {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
                           gcn_args={"units": 64, "use_bias": True, "activation": "relu", "pooling_method": "sum"},
                           depth=3, output_embedding="node",
                           output_mlp={"use_bias": [True, True, False], "units": [64, 32, 7],
                                       "activation": ["relu", "relu", "softmax"]})
    t_layers = timeit(lambda: model(ins, fused=False), 50)
    model(ins), model(ins)
    assert model.fused is not None and model.fused.last == "graph"
    t_graph = timeit(lambda: model(ins), 500)
    out = {"workload": "BASELINE config 5: GCN.make_model on one Cora-shaped graph, N=%d, M=%d (incl. self loops), F=%d" %
                       (n, m, f), "nodes": n, "edges": m, "forward_ms_layer_path_eager": t_layers * 1e3,
           "forward_ms_fused_graph_replay": t_graph * 1e3, "edges_per_s": m / t_graph}
    timer = _HipTimer()
    # the fused route's launches alone (descriptors of the bound slot)
    import ctypes
    slot = model.fused.slot_of(ins)
    descs = slot._descs(torch.empty((n, 7), device="cuda"))
    for name, d, alg, flops in (
            ("gcn_tile_kernel<input> X W0 + b0, then gcn0 Dense", descs[0], 4 * (n * f + f * 64 + n * 64),
             2.0 * n * f * 64 + 2.0 * n * 64 * 64),
            ("gcn_tile_kernel<aggregate> sum_e w_e h[send] -> relu -> next Dense", descs[1], 20 * m + 8 * n * 64,
             2.0 * m * 64 + 2.0 * n * 64 * 64)):
        ms = timer.time_ms(lambda: _ffi.call("mp_gcn_tile_f32", ctypes.byref(d), _ffi.stream()), 50)
        out[name] = {"avg_launch_us": ms * 1e3, "algorithmic_bytes": alg, "gbs": alg / (ms * 1e-3) / 1e9,
                     "bound": "hbm", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK, "tflops": flops / (ms * 1e-3) / 1e12}
    x = ins[0].values
    w = torch.randn(f, 64, device="cuda") * 0.05
    b = torch.zeros(64, device="cuda")
    y = torch.empty(n, 64, device="cuda")
    tiles = -(-n // 64)
    splits = max(1, min(256 // tiles, f // 128, 64))
    ws = torch.empty((splits, n, 64), device="cuda")
    alg = 4 * (n * f + f * 64 + n * 64)
    for name, fn in (("dense_mfma_kernel split-K x%d (2708,1433)x(1433,64)" % splits,
                      lambda: _ffi.call("mp_dense_splitk_f32", _ffi.ptr(x), n, f, _ffi.ptr(w), _ffi.ptr(b), 64, 0, 0.0, splits,
                                        _ffi.ptr(ws), ws.numel() * 4, _ffi.ptr(y), _ffi.stream())),
                     ("dense_mfma_kernel one k range (43 workgroups)",
                      lambda: _ffi.call("mp_dense_f32", _ffi.ptr(x), n, f, _ffi.ptr(w), _ffi.ptr(b), 64, 0, 0.0, _ffi.ptr(y),
                                        _ffi.stream()))):
        ms = timer.time_ms(fn, 50)
        out[name] = {"avg_launch_us": ms * 1e3, "algorithmic_bytes": alg, "gbs": alg / (ms * 1e-3) / 1e9,
                     "bound": "hbm", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK,
                     "tflops": 2.0 * n * f * 64 / (ms * 1e-3) / 1e12}
    plan = ins[2].index_plan(ins[0])
    ptr, perm, _ = plan.csr(0)
    h = torch.randn(n, 64, device="cuda")
    agg = torch.empty(n, 64, device="cuda")
    wts = ins[1].values.contiguous().view(-1)
    ms = timer.time_ms(lambda: _ffi.call("mp_gather_segment_reduce_csr_f32", _ffi.MP_SUM, _ffi.ptr(h), n, 64,
                                         _ffi.ptr(plan.col(1).contiguous()), m, _ffi.ptr(ptr), _ffi.ptr(perm), n,
                                         _ffi.ptr(wts), 0, 1, 0.05, _ffi.ptr(agg), _ffi.stream()), 50)
    alg = 20 * m + 8 * n * 64
    out["gather_segment_reduce_csr_kernel (GCN aggregate)"] = {
        "avg_launch_us": ms * 1e3, "algorithmic_bytes": alg, "gbs": alg / (ms * 1e-3) / 1e9, "bound": "hbm",
        "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK, "max_in_degree": int((ptr[1:] - ptr[:-1]).max().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
