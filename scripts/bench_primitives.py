"""HBM-roofline microbenchmarks of the scatter-gather primitives (HIP events, algorithmic bytes of SURVEY.md 8d), plus
forward times of the PaiNN / GCN configs.  Writes a markdown table to stdout.

    python scripts/bench_primitives.py [graphs]      # default 12500 QM9-shaped graphs (config-4 shard size)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.engine import _HipTimer
from gcnn_keras_amd.layers.gather import GatherNodes, GatherNodesOutgoing
from gcnn_keras_amd.layers.pooling import PoolingLocalEdges, PoolingNodes
from gcnn_keras_amd.ragged import RaggedTensor

HBM = 8000.0


def main():
    graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
    b = synth.qm9_like_batch(num_graphs=graphs, seed=1234)
    n, m, f = int(b["node_splits"][-1]), int(b["edge_splits"][-1]), 128
    rng = np.random.default_rng(0)
    nodes = RaggedTensor.from_numpy(rng.normal(size=(n, f)).astype(np.float32), b["node_splits"])
    edges = RaggedTensor.from_numpy(rng.normal(size=(m, f)).astype(np.float32), b["edge_splits"])
    idx = RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])
    w = torch.rand(m, device="cuda")
    plan = idx.index_plan(nodes)
    ptr, perm, _ = plan.csr(0)
    timer = _HipTimer()
    rows = []

    def bench(name, fn, nbytes, iters=20):
        ms = timer.time_ms(fn, iters)
        gbs = nbytes / (ms * 1e-3) / 1e9
        rows.append((name, ms * 1e3, nbytes / 1e6, gbs, gbs / HBM))

    g_out, g_cat, pool, pooln = GatherNodesOutgoing(), GatherNodes(), PoolingLocalEdges("sum"), PoolingNodes("sum")
    cols_buf = torch.empty((2, m), dtype=torch.int32, device="cuda")
    flag_buf = torch.zeros(1, dtype=torch.int32, device="cuda")
    bench("index plan (mp_index_prepare_i64)",
          lambda: _ffi.call("mp_index_prepare_i64", _ffi.ptr(idx.values), m, 2, _ffi.ptr(nodes.row_splits),
                            _ffi.ptr(idx.row_splits), graphs, n, _ffi.ptr(cols_buf), _ffi.ptr(flag_buf), _ffi.stream()),
          16 * m + 8 * m)
    bench("GatherNodesOutgoing", lambda: g_out([nodes, idx]), 8 * m + 4 * n * f + 4 * m * f)
    bench("GatherNodes (concat i||j)", lambda: g_cat([nodes, idx]), 16 * m + 4 * n * f + 8 * m * f)
    bench("PoolingLocalEdges(sum)", lambda: pool([nodes, edges, idx]), 4 * m * f + 8 * m + 4 * n * f)
    bench("PoolingNodes(sum)", lambda: pooln(nodes), 4 * n * f + 8 * (graphs + 1) + 4 * graphs * f)
    out = torch.empty((n, f), device="cuda")
    send = plan.col(1).contiguous()
    bench("GCN aggregate fused (gather*w -> sum, F=128)",
          lambda: _ffi.call("mp_gather_segment_reduce_csr_f32", 0, _ffi.ptr(nodes.values), n, f, _ffi.ptr(send), m,
                            _ffi.ptr(ptr), _ffi.ptr(perm), n, _ffi.ptr(w), 0, 1, 0.0, _ffi.ptr(out), _ffi.stream()),
          20 * m + 8 * n * f)
    print("| primitive (N=%d, M=%d, F=%d) | us | algorithmic MB | GB/s | of 8 TB/s |" % (n, m, f))
    print("|---|---|---|---|---|")
    for r in rows:
        print("| %s | %.1f | %.1f | %.0f | %.1f %% |" % (r[0], r[1], r[2], r[3], 100 * r[4]))

    # model forwards (layer path): BASELINE configs 3 and 5
    from gcnn_keras_amd.literature import GCN, PAiNN
    from gcnn_keras_amd.model.force import EnergyForceModel
    bp = synth.md17_like_batch()
    pm = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
    inp = [RaggedTensor.from_numpy(bp["node_number"], bp["node_splits"]),
           RaggedTensor.from_numpy(bp["node_coordinates"], bp["node_splits"]),
           RaggedTensor.from_numpy(bp["edge_indices"], bp["edge_splits"])]
    ms = timer.time_ms(lambda: pm(inp), 10)
    print("\nPaiNN forward (config 3: 64 graphs, N=%d, M=%d), layer path: %.2f ms = %.1f M edges/s"
          % (bp["node_splits"][-1], bp["edge_splits"][-1], ms, bp["edge_splits"][-1] / ms / 1e3))
    from gcnn_keras_amd.engine import GraphedModel
    gp = GraphedModel(pm, inp)
    ref = pm(inp).cpu().numpy()
    assert np.max(np.abs(gp().cpu().numpy() - ref)) <= 1e-6 * np.max(np.abs(ref))
    ms = timer.time_ms(lambda: gp(), 20)
    print("PaiNN forward, same layer path replayed from one HIP graph (GraphedModel): %.3f ms = %.1f M edges/s"
          % (ms, bp["edge_splits"][-1] / ms / 1e3))
    efm = EnergyForceModel(model_energy=pm, energy_output=0, output_squeeze_states=True)
    ms = timer.time_ms(lambda: efm(inp), 5)
    print("PaiNN energy + force (EnergyForceModel), layer path forward + reverse: %.2f ms" % ms)
    g = synth.cora_like_graph()
    gm = GCN.make_model(
        inputs=[{"shape": (None, 1433), "name": "node_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, 1), "name": "edge_weights", "dtype": "float32", "ragged": True},
                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
        gcn_args={"units": 64, "use_bias": True, "activation": "relu", "pooling_method": "sum"}, depth=3,
        output_embedding="node", output_to_tensor=False,
        output_mlp={"use_bias": [True, True, True], "units": [64, 32, 7], "activation": ["relu", "relu", "softmax"]})
    ginp = [RaggedTensor.from_numpy(g["node_attributes"], g["node_splits"]),
            RaggedTensor.from_numpy(g["edge_weights"], g["edge_splits"]),
            RaggedTensor.from_numpy(g["edge_indices"], g["edge_splits"])]
    ms = timer.time_ms(lambda: gm(ginp), 10)
    gg = GraphedModel(gm, ginp)
    ms_g = timer.time_ms(lambda: gg(), 20)
    print("GCN forward replayed from one HIP graph: %.3f ms = %.1f M edges/s" % (ms_g, g["edge_splits"][-1] / ms_g / 1e3))
    deg = np.bincount(g["edge_indices"][:, 0]).max()
    print("GCN forward (config 5: N=2708, M=%d incl. self loops, max in-degree %d), layer path: %.3f ms = %.1f M edges/s"
          % (g["edge_splits"][-1], deg, ms, g["edge_splits"][-1] / ms / 1e3))


if __name__ == "__main__":
    main()
