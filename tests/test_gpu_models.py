"""PaiNN and GCN forwards (layer path on the HIP engine) and MessagePassingBase vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth
from oracle import kgcnn_oracle as ko
from parity import assert_rows_close

pytestmark = pytest.mark.gpu


def _dev(values, splits):
    from gcnn_keras_amd.ragged import RaggedTensor
    return RaggedTensor.from_numpy(values, splits)


def _check(got, ref32, ref64, tol=1e-5):
    scale = float(np.max(np.abs(ref64)))
    assert got.shape == ref32.shape
    assert np.max(np.abs(got - ref32)) <= tol * scale, (np.max(np.abs(got - ref32)), scale)
    err_e, err_o = float(np.max(np.abs(got - ref64))), float(np.max(np.abs(ref32 - ref64)))
    assert err_e <= max(4 * err_o, 2e-6 * scale), (err_e, err_o, scale)


@pytest.mark.parametrize("num_graphs,seed,method,cutoff", [(3, 12, "eps", None), (64, 2345, "eps", None),
                                                           (4, 7, "zeros", 5.0)])
def test_painn_forward(num_graphs, seed, method, cutoff, golden_dir):
    """BASELINE config 3 shape: PAiNN.make_model defaults, Bessel(20, 5.0, 5), equivariant init 'eps'."""
    from gcnn_keras_amd.literature import PAiNN
    b = synth.md17_like_batch(num_graphs=num_graphs, seed=seed)
    p = synth.painn_params(seed=8, random_bias=True)
    model = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": method},
                             conv_args={"units": 128, "cutoff": cutoff, "conv_pool": "sum"})
    names = [n for n, _ in model.weights]
    assert len(names) == len(p), (len(names), len(p))
    # constructor order: embedding, bessel frequencies, then per block conv (dense1, phi, w) and update
    order = ["embedding", "bessel/frequencies"]
    for i in range(3):
        order += ["conv%d/dense1/kernel" % i, "conv%d/dense1/bias" % i, "conv%d/phi/kernel" % i, "conv%d/phi/bias" % i,
                  "conv%d/w/kernel" % i, "conv%d/w/bias" % i,
                  "update%d/dense1/kernel" % i, "update%d/dense1/bias" % i, "update%d/lin_u/kernel" % i,
                  "update%d/lin_v/kernel" % i, "update%d/a/kernel" % i, "update%d/a/bias" % i]
    order += ["output_mlp/0/kernel", "output_mlp/0/bias", "output_mlp/1/kernel", "output_mlp/1/bias"]
    model.set_weights([p[k] for k in order])
    out = model([_dev(b["node_number"], b["node_splits"]), _dev(b["node_coordinates"], b["node_splits"]),
                 _dev(b["edge_indices"], b["edge_splits"])]).cpu().numpy()

    def oracle(dtype):
        return ko.painn_forward(ko.to_dtype(p, dtype), ko.R(b["node_number"], b["node_splits"]),
                                ko.R(b["node_coordinates"].astype(dtype), b["node_splits"]),
                                ko.R(b["edge_indices"], b["edge_splits"]), depth=3, equiv_method=method, cutoff=cutoff)
    _check(out, oracle(np.float32), oracle(np.float64))
    if (num_graphs, seed) == (3, 12):
        frozen = np.load(os.path.join(golden_dir, "frozen_painn_small.npz"))["out"]
        assert np.max(np.abs(out - frozen)) <= 1e-5 * np.max(np.abs(frozen))


@pytest.mark.parametrize("case", ["small", "cora"])
def test_gcn_forward(case, golden_dir):
    """BASELINE config 5 shape: GCN.make_model, units 64, depth 3, node output [64, 32, 7] softmax."""
    from gcnn_keras_amd.literature import GCN
    if case == "small":
        g = synth.cora_like_graph(num_nodes=120, num_features=40, seed=13, drop_pairs=9)
        feats = 40
    else:
        g = synth.cora_like_graph()
        feats = 1433
        assert g["edge_splits"][-1] == 10556 + 2708
    p = synth.gcn_params(seed=9, in_features=feats, random_bias=True)
    model = GCN.make_model(
        inputs=[{"shape": (None, feats), "name": "node_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, 1), "name": "edge_weights", "dtype": "float32", "ragged": True},
                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
        gcn_args={"units": 64, "use_bias": True, "activation": "relu", "pooling_method": "sum"},
        depth=3, output_embedding="node", output_to_tensor=False,
        output_mlp={"use_bias": [True, True, True], "units": [64, 32, 7], "activation": ["relu", "relu", "softmax"]})
    model.set_weights(list(p.values()))
    out = model([_dev(g["node_attributes"], g["node_splits"]), _dev(g["edge_weights"], g["edge_splits"]),
                 _dev(g["edge_indices"], g["edge_splits"])]).values.cpu().numpy()

    def oracle(dtype):
        return ko.gcn_forward(ko.to_dtype(p, dtype), ko.R(g["node_attributes"].astype(dtype), g["node_splits"]),
                              ko.R(g["edge_weights"].astype(dtype), g["edge_splits"]),
                              ko.R(g["edge_indices"], g["edge_splits"])).values
    _check(out, oracle(np.float32), oracle(np.float64))
    assert np.allclose(out.sum(-1), 1.0, atol=1e-5)
    if case == "small":
        frozen = np.load(os.path.join(golden_dir, "frozen_gcn_small.npz"))["out"]
        assert np.max(np.abs(out - frozen)) <= 1e-5 * np.max(np.abs(frozen))


def test_message_passing_base():
    """The README API example pattern (reference README.md:118-138): subclass with message / update functions."""
    from gcnn_keras_amd.layers.message import MessagePassingBase
    from gcnn_keras_amd.layers.modules import Dense, LazyAdd, LazyConcatenate

    class MyMessageNN(MessagePassingBase):
        def __init__(self, units, **kwargs):
            super().__init__(**kwargs)
            self.dense = Dense(units)
            self.add = LazyAdd()
            self.cat = LazyConcatenate(axis=-1)

        def message_function(self, inputs, **kwargs):
            n_in, n_out, edges = inputs
            return self.dense(self.cat([n_in, n_out]))

        def update_nodes(self, inputs, **kwargs):
            nodes, nodes_update = inputs
            return self.add([nodes, nodes_update])

    t = synth.toy_batch()
    n = ko.R(np.tile(t["node_attributes"], (1, 1)), t["node_splits"])
    ei = ko.R(t["edge_indices"], t["edge_splits"])
    lay = MyMessageNN(3)
    out = lay([_dev(n.values, n.row_splits), _dev(np.zeros((6, 1), np.float32), t["edge_splits"]),
               _dev(ei.values, ei.row_splits)])
    w, bias = lay.dense.get_weights()
    g = ko.gather_nodes_selection(n, ei, [0, 1])
    msg = ko.dense(ko.lazy_concatenate(g), w, bias, None)
    ref = ko.lazy_add([n, ko.pooling_local_edges(n, msg, ei, "sum")])
    assert np.max(np.abs(out.values.cpu().numpy() - ref.values)) <= 1e-5 * np.max(np.abs(ref.values))
    assert lay.get_config()["pooling_method"] == "sum"


@pytest.mark.parametrize("method,normalize,act", [("sum", False, "relu"), ("mean", False, "linear"),
                                                   ("sum", True, "kgcnn>leaky_relu"), ("max", False, "relu")])
def test_gcn_layer_fused_aggregate_equals_layer_sequence(method, normalize, act):
    """GCN.call's fused gather+weighted-pool+activation kernel vs the oracle's layer sequence (gcn_conv.py:85-90),
    on an unsorted edge list."""
    from gcnn_keras_amd.layers.conv.gcn_conv import GCN
    g = synth.cora_like_graph(num_nodes=90, num_features=24, seed=3, drop_pairs=5)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(g["edge_indices"]))
    ei, ew = g["edge_indices"][perm], g["edge_weights"][perm]
    k, bias = synth.glorot_uniform(rng, 24, 32), rng.uniform(-.1, .1, 32).astype(np.float32)
    lay = GCN(units=32, pooling_method=method, normalize_by_weights=normalize, activation=act)
    lay.ensure_built([(None, None, 24), (None, None, 1), (None, None, 2)])
    lay.set_weights([k, bias])
    out = lay([_dev(g["node_attributes"], g["node_splits"]), _dev(ew, g["edge_splits"]), _dev(ei, g["edge_splits"])])
    ref = ko.gcn_layer(ko.R(g["node_attributes"], g["node_splits"]), ko.R(ew, g["edge_splits"]),
                       ko.R(ei, g["edge_splits"]), {"kernel": k, "bias": bias}, act=act, pooling_method=method,
                       normalize_by_weights=normalize)
    got = out.values.cpu().numpy()
    assert got.shape == ref.values.shape
    assert np.max(np.abs(got - ref.values)) <= 1e-5 * max(np.max(np.abs(ref.values)), 1e-30)


def test_graphed_model_replay_matches_eager():
    """engine.GraphedModel: the layer path captured once into a HIP graph replays to the same numbers, and follows
    in-place updates of the input values."""
    from gcnn_keras_amd.engine import GraphedModel
    from gcnn_keras_amd.literature import Schnet
    b = synth.qm9_like_batch(num_graphs=5, seed=31)
    p = synth.schnet_params(seed=7, random_bias=True)
    model = Schnet.make_model(depth=2)
    model.set_weights(list(synth.schnet_params(seed=7, depth=2, random_bias=True).values()))
    inputs = [_dev(b["node_number"], b["node_splits"]), _dev(b["node_coordinates"], b["node_splits"]),
              _dev(b["edge_indices"], b["edge_splits"])]
    eager = model(inputs).cpu().numpy()
    gm = GraphedModel(model, inputs)
    assert np.array_equal(gm().cpu().numpy(), eager)
    inputs[1].values.mul_(1.01)          # new coordinates, same topology
    assert np.array_equal(gm().cpu().numpy(), model(inputs).cpu().numpy())
    assert not np.array_equal(gm().cpu().numpy(), eager)


@pytest.mark.parametrize("cutoff,shuffle", [(None, False), (5.0, True)])
def test_painn_conv_fused_message_vs_oracle(cutoff, shuffle):
    """PAiNNconv.call with the fused edge kernel vs the oracle's layer sequence (painn_conv.py:97-115)."""
    from gcnn_keras_amd.layers.conv.painn_conv import PAiNNconv
    b = synth.md17_like_batch(num_graphs=3, seed=4)
    rng = np.random.default_rng(1)
    idx = b["edge_indices"].copy()
    if shuffle:
        for g in range(3):
            lo, hi = b["edge_splits"][g], b["edge_splits"][g + 1]
            idx[lo:hi] = idx[lo:hi][rng.permutation(hi - lo)]
    n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
    z = ko.R(rng.normal(size=(n, 128)).astype(np.float32), b["node_splits"])
    v = ko.R(rng.normal(size=(n, 3, 128)).astype(np.float32), b["node_splits"])
    ridx = ko.R(idx, b["edge_splits"])
    xyz = ko.R(b["node_coordinates"], b["node_splits"])
    p1, p2 = ko.node_position(xyz, ridx)
    d = ko.node_distance_euclidean(p1, p2)
    rij = ko.edge_direction_normalized(p1, p2)
    rbf = ko.bessel_basis(d, 20, 5.0)
    env = ko.cos_cutoff_envelope(d, cutoff)
    p = {k[len("conv0/"):]: v_ for k, v_ in synth.painn_params(seed=8, random_bias=True).items()
         if k.startswith("conv0/")}
    lay = PAiNNconv(units=128, cutoff=cutoff)
    lay.ensure_built([(None, None, 128), (None, None, 3, 128), (None, None, 20), (None, None, 1), (None, None, 3),
                      (None, None, 2)])
    lay.set_weights([p["dense1/kernel"], p["dense1/bias"], p["phi/kernel"], p["phi/bias"], p["w/kernel"], p["w/bias"]])
    ds, dv = lay([_dev(z.values, z.row_splits), _dev(v.values, v.row_splits), _dev(rbf.values, rbf.row_splits),
                  _dev(env.values, env.row_splits), _dev(rij.values, rij.row_splits), _dev(idx, b["edge_splits"])])
    rds, rdv = ko.painn_conv(z, v, rbf, env, rij, ridx, p, cutoff=cutoff)
    for got, ref in ((ds.values.cpu().numpy(), rds.values), (dv.values.cpu().numpy(), rdv.values)):
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref)) <= 1e-5 * np.max(np.abs(ref))


def test_graphed_model_pool_overlaps_batches_of_a_layer_path_model():
    """``GraphedModelPool``: three PaiNN batches in flight (own inputs / graph / stream each) return what the eager
    model returns for each batch."""
    import time
    from gcnn_keras_amd.engine import GraphedModelPool
    from gcnn_keras_amd.literature import PAiNN
    from gcnn_keras_amd.ragged import RaggedTensor
    model = PAiNN.make_model()
    batches = [synth.md17_like_batch(num_graphs=8, seed=40 + k) for k in range(3)]
    inputs = [[RaggedTensor.from_numpy(b["node_number"], b["node_splits"]),
               RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
               RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])] for b in batches]
    eager = [model(x).clone() for x in inputs]
    pool = GraphedModelPool(model, inputs)
    for i in range(12):
        pool.replay(i)
    torch.cuda.synchronize()
    for k in range(3):
        assert torch.equal(pool.slots[k].output, eager[k])
    t0 = time.perf_counter()
    for i in range(60):
        pool.replay(i)
    torch.cuda.synchronize()
    t_pool = (time.perf_counter() - t0) / 60
    one = pool.slots[0]
    t0 = time.perf_counter()
    for i in range(60):
        one()
    torch.cuda.synchronize()
    t_one = (time.perf_counter() - t0) / 60
    print("PaiNN forward (8 graphs): one graph at a time %.0f us, three in flight %.0f us per forward" % (t_one * 1e6, t_pool * 1e6))


def test_graphed_models_outlive_the_routes_slot_table_and_follow_weight_updates():
    """A GraphedModel of a fused-route model keeps the batch slot its graph addresses alive: ten captured batches on a
    route with max_slots = 8 (the first captures are evicted from the route's table), then ``route.release()`` - every
    graph still replays the oracle's numbers.  ``set_weights`` after the capture is picked up by the replay (derived
    weight images re-filled in place); replacing a weight tensor object makes the replay raise."""
    from gcnn_keras_amd import _ffi
    from gcnn_keras_amd.engine import GraphedModel
    from gcnn_keras_amd.literature import Schnet
    p = synth.schnet_params(seed=7, depth=2, random_bias=True)
    model = Schnet.make_model(depth=2)
    model.set_weights(list(p.values()))
    assert model.fused is not None and model.fused.max_slots == 8
    batches = [synth.qm9_like_batch(num_graphs=4, seed=60 + k) for k in range(10)]
    inputs = [[_dev(b["node_number"], b["node_splits"]), _dev(b["node_coordinates"], b["node_splits"]),
               _dev(b["edge_indices"], b["edge_splits"])] for b in batches]

    def oracle(pp, b):
        return ko.schnet_forward(pp, ko.R(b["node_number"], b["node_splits"]), ko.R(b["node_coordinates"], b["node_splits"]),
                                 ko.R(b["edge_indices"], b["edge_splits"]), depth=2)

    graphed = [GraphedModel(model, x) for x in inputs]
    assert len(model.fused._slots) <= 8
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]   # re-use freed blocks, if any were freed
    for g, b in zip(graphed, batches):
        assert_rows_close(g().cpu().numpy(), oracle(p, b), what="graphed batch beyond max_slots")
    model.fused.release()
    junk += [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]
    for g, b in zip(graphed, batches):
        assert_rows_close(g().cpu().numpy(), oracle(p, b), what="graphed batch after release()")
    p2 = synth.schnet_params(seed=11, depth=2, random_bias=True)
    model.set_weights(list(p2.values()))
    for g, b in zip(graphed[:3], batches[:3]):
        assert_rows_close(g().cpu().numpy(), oracle(p2, b), what="graphed batch after set_weights")
    lay = next(l for l in model.layers if torch.is_tensor(getattr(l, "kernel", None)))
    lay.kernel = lay.kernel.clone()
    with pytest.raises(_ffi.EngineError):
        graphed[0]()
    del junk


@pytest.mark.parametrize("v2,concat", [(False, False), (False, True), (True, False)])
def test_gat_builders_forward(v2, concat):
    """``GAT.make_model`` / ``GATv2.make_model`` (kgcnn/literature/GAT.py:89-121) with feature inputs vs the oracle."""
    from gcnn_keras_amd.literature import GAT, GATv2
    from gcnn_keras_amd.ragged import RaggedTensor
    b = synth.qm9_like_batch(num_graphs=6, seed=17)
    rng = np.random.default_rng(18)
    n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
    fn, fe, units, heads, depth = 12, 6, 16, 3, 2
    x = rng.normal(size=(n, fn)).astype(np.float32)
    e = rng.normal(size=(m, fe)).astype(np.float32)
    builder = GATv2 if v2 else GAT
    model = builder.make_model(
        inputs=[{"shape": (None, fn), "name": "node_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, fe), "name": "edge_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
        attention_args={"units": units}, depth=depth, attention_heads_num=heads, attention_heads_concat=concat)
    # random weights in model order -> oracle parameter dict
    p, arrays = {}, []
    def add(key, shape, bias=False):
        a = (rng.normal(size=shape) * 0.1).astype(np.float32) if bias else synth.glorot_uniform(rng, shape[0], shape[1])
        p[key] = a
        arrays.append(a)
    arrays += [w.cpu().numpy() for _, w in model.layers[0].weights] + [w.cpu().numpy() for _, w in model.layers[1].weights]
    add("dense0/kernel", (fn, units)); add("dense0/bias", (units,), True)
    width = units
    for i in range(depth):
        for h in range(heads):
            pre = "block%d/head%d/" % (i, h)
            add(pre + "linear_trafo/kernel", (width, units)); add(pre + "linear_trafo/bias", (units,), True)
            if v2:
                add(pre + "alpha_activation/kernel", (2 * width + fe, units))
                add(pre + "alpha_activation/bias", (units,), True)
                add(pre + "alpha/kernel", (units, 1))
            else:
                add(pre + "alpha/kernel", (2 * units + fe, 1))
        width = units * heads if concat else units
    for k, (fin, fout, has_bias) in enumerate(((width, 25, True), (25, 10, True), (10, 1, False))):
        add("output_mlp/%d/kernel" % k, (fin, fout))
        if has_bias:
            add("output_mlp/%d/bias" % k, (fout,), True)
    model.set_weights(arrays)
    out = model([RaggedTensor.from_numpy(x, b["node_splits"]), RaggedTensor.from_numpy(e, b["edge_splits"]),
                 RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]).cpu().numpy()
    ref = ko.gat_forward(p, ko.R(x, b["node_splits"]), ko.R(e, b["edge_splits"]), ko.R(b["edge_indices"], b["edge_splits"]),
                         depth=depth, heads=heads, concat_heads=concat, v2=v2)
    assert out.shape == (6, 1)
    assert np.max(np.abs(out - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref)))
    with pytest.raises(ValueError):
        builder.make_model(not_a_kwarg=1)


def test_dmpnn_builder_forward():
    """``DMPNN.make_model`` (kgcnn/literature/DMPNN.py:119-171) with feature inputs vs the oracle."""
    from gcnn_keras_amd.literature import DMPNN
    from gcnn_keras_amd.ragged import RaggedTensor
    b = synth.qm9_like_batch(num_graphs=5, seed=23)
    rng = np.random.default_rng(24)
    n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
    fn, fe, units, depth = 10, 6, 32, 3
    x = rng.normal(size=(n, fn)).astype(np.float32)
    e = rng.normal(size=(m, fe)).astype(np.float32)
    pairs = np.full((m, 1), -1, dtype=np.int64)      # position of the reverse edge inside its graph (SetRange graphs
    es = b["edge_splits"]                            # are symmetric, so every edge has one)
    for g in range(5):
        seg = b["edge_indices"][es[g]:es[g + 1]]
        pos = {(int(i), int(j)): k for k, (i, j) in enumerate(seg)}
        for k, (i, j) in enumerate(seg):
            pairs[es[g] + k, 0] = pos.get((int(j), int(i)), -1)
    assert (pairs >= 0).all()
    model = DMPNN.make_model(
        inputs=[{"shape": (None, fn), "name": "node_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, fe), "name": "edge_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True},
                {"shape": (None, 1), "name": "edge_indices_reverse", "dtype": "int64", "ragged": True}],
        edge_initialize={"units": units}, edge_dense={"units": units}, node_dense={"units": units}, depth=depth,
        output_mlp={"units": [16, 8, 1]})
    p, arrays = {}, []
    arrays += [w.cpu().numpy() for lay in model.layers[:2] for _, w in lay.weights]     # embeddings (unused here)
    def add(key, fin, fout):
        p[key + "/kernel"] = synth.glorot_uniform(rng, fin, fout)
        p[key + "/bias"] = (rng.normal(size=fout) * 0.1).astype(np.float32)
        arrays.extend([p[key + "/kernel"], p[key + "/bias"]])
    add("h0", fn + fe, units); add("edge", units, units); add("node", units + fn, units)
    for k, (fin, fout, has_bias) in enumerate(((units, 16, True), (16, 8, True), (8, 1, False))):
        p["output_mlp/%d/kernel" % k] = synth.glorot_uniform(rng, fin, fout)
        arrays.append(p["output_mlp/%d/kernel" % k])
        if has_bias:
            p["output_mlp/%d/bias" % k] = (rng.normal(size=fout) * 0.1).astype(np.float32)
            arrays.append(p["output_mlp/%d/bias" % k])
    model.set_weights(arrays)
    out = model([RaggedTensor.from_numpy(x, b["node_splits"]), RaggedTensor.from_numpy(e, es),
                 RaggedTensor.from_numpy(b["edge_indices"], es), RaggedTensor.from_numpy(pairs, es)]).cpu().numpy()
    ref = ko.dmpnn_forward(p, ko.R(x, b["node_splits"]), ko.R(e, es), ko.R(b["edge_indices"], es), ko.R(pairs, es),
                           depth=depth)
    assert out.shape == (5, 1)
    assert np.max(np.abs(out - ref)) <= 2e-5 * max(1.0, np.max(np.abs(ref)))
    with pytest.raises(NotImplementedError):
        DMPNN.make_model(use_graph_state=True)


def test_graphsage_builder_forward():
    """``GraphSAGE.make_model`` (kgcnn/literature/GraphSAGE.py:95-135) with feature inputs vs the oracle."""
    from gcnn_keras_amd.literature import GraphSAGE
    from gcnn_keras_amd.ragged import RaggedTensor
    b = synth.qm9_like_batch(num_graphs=6, seed=27)
    rng = np.random.default_rng(28)
    n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
    fn, fe, depth = 10, 6, 2
    x = rng.normal(size=(n, fn)).astype(np.float32)
    e = rng.normal(size=(m, fe)).astype(np.float32)
    model = GraphSAGE.make_model(
        inputs=[{"shape": (None, fn), "name": "node_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, fe), "name": "edge_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
        node_mlp_args={"units": [24, 12]}, edge_mlp_args={"units": [20, 14]}, depth=depth)
    p, arrays = {}, []
    arrays += [w.cpu().numpy() for lay in model.layers[:2] for _, w in lay.weights]
    def dense_pair(key, fin, fout, bias=True):
        p[key + "/kernel"] = synth.glorot_uniform(rng, fin, fout)
        arrays.append(p[key + "/kernel"])
        if bias:
            p[key + "/bias"] = (rng.normal(size=fout) * 0.1).astype(np.float32)
            arrays.append(p[key + "/bias"])
    width = fn
    for i in range(depth):
        dense_pair("block%d/edge/0" % i, width + fe, 20); dense_pair("block%d/edge/1" % i, 20, 14)
        dense_pair("block%d/node/0" % i, width + 14, 24); dense_pair("block%d/node/1" % i, 24, 12)
        p["block%d/norm/gamma" % i] = rng.uniform(0.5, 1.5, 12).astype(np.float32)
        p["block%d/norm/beta" % i] = rng.normal(size=12).astype(np.float32)
        arrays += [p["block%d/norm/gamma" % i], p["block%d/norm/beta" % i]]
        width = 12
    dense_pair("output_mlp/0", 12, 25); dense_pair("output_mlp/1", 25, 10); dense_pair("output_mlp/2", 10, 1, bias=False)
    model.set_weights(arrays)
    out = model([RaggedTensor.from_numpy(x, b["node_splits"]), RaggedTensor.from_numpy(e, b["edge_splits"]),
                 RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]).cpu().numpy()
    ref = ko.graphsage_forward(p, ko.R(x, b["node_splits"]), ko.R(e, b["edge_splits"]),
                               ko.R(b["edge_indices"], b["edge_splits"]), depth=depth)
    assert out.shape == (6, 1)
    assert np.max(np.abs(out - ref)) <= 3e-5 * max(1.0, np.max(np.abs(ref)))
    with pytest.raises(NotImplementedError):
        GraphSAGE.make_model(pooling_args={"pooling_method": "LSTM"})


def test_gin_builder_forward():
    """``GIN.make_model`` (kgcnn/literature/GIN.py:82-116) with feature inputs, no normalisation, vs the oracle."""
    from gcnn_keras_amd.literature import GIN
    from gcnn_keras_amd.ragged import RaggedTensor
    b = synth.qm9_like_batch(num_graphs=6, seed=29)
    rng = np.random.default_rng(30)
    n = int(b["node_splits"][-1])
    fn, units, depth, classes = 9, 16, 2, 3
    x = rng.normal(size=(n, fn)).astype(np.float32)
    model = GIN.make_model(
        inputs=[{"shape": (None, fn), "name": "node_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
        gin_mlp={"units": [units, units]}, gin_args={"epsilon_learnable": True}, depth=depth,
        last_mlp={"units": [12, 12, 8]}, output_mlp={"units": classes})
    p, arrays = {}, []
    arrays += [w.cpu().numpy() for _, w in model.layers[0].weights]
    def dense_pair(key, fin, fout):
        p[key + "/kernel"] = synth.glorot_uniform(rng, fin, fout)
        p[key + "/bias"] = (rng.normal(size=fout) * 0.1).astype(np.float32)
        arrays.extend([p[key + "/kernel"], p[key + "/bias"]])
    dense_pair("dense0", fn, units)
    for i in range(depth):
        p["gin%d/eps" % i] = np.float32(0.1 * (i + 1))
        arrays.append(p["gin%d/eps" % i])
        dense_pair("mlp%d/0" % i, units, units); dense_pair("mlp%d/1" % i, units, units)
    for j in range(depth + 1):
        dense_pair("last%d/0" % j, units, 12); dense_pair("last%d/1" % j, 12, 12); dense_pair("last%d/2" % j, 12, 8)
    dense_pair("output", 8, classes)
    model.set_weights(arrays)
    out = model([RaggedTensor.from_numpy(x, b["node_splits"]),
                 RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]).cpu().numpy()
    ref = ko.gin_forward(p, ko.R(x, b["node_splits"]), ko.R(b["edge_indices"], b["edge_splits"]), depth=depth)
    assert out.shape == (6, classes) and np.allclose(out.sum(axis=1), 1.0, atol=1e-5)
    assert np.max(np.abs(out - ref)) <= 2e-5
    with pytest.raises(NotImplementedError):
        GIN.make_model(gin_mlp={"use_normalization": True})


@pytest.mark.parametrize("units,mlp_units,mlp_act,sort_edges", [(64, [64, 32, 7], ["relu", "relu", "softmax"], True),
                                                               (32, [16, 5], ["kgcnn>leaky_relu", "linear"], False),
                                                               (128, [100], ["softmax"], True)])
def test_gcn_fused_route_equals_layer_sequence(units, mlp_units, mlp_act, sort_edges):
    """``GCN.make_model(...)(inputs)`` takes the tile kernels of csrc/mp_gcn.hip (1 + depth launches, graph-replayed for
    a re-bound input set); ``fused=False`` runs the reference's layer sequence on the same weights.  Several graphs of
    different sizes, a hub receiver (in-degree > 100: its edges are shared by all thread groups of a tile), an isolated
    node, a feature count that is not a multiple of 16, optionally an edge list that is not receiver-sorted."""
    from gcnn_keras_amd import _ffi
    from gcnn_keras_amd.literature import GCN
    rng = np.random.default_rng(units)
    sizes = [37, 150, 16, 1, 60]
    feats = 45
    attrs, wts, idxs, ns, es = [], [], [], [0], [0]
    for g, n in enumerate(sizes):
        m = 0 if n == 1 else 4 * n
        i = rng.integers(0, n, size=m)
        j = rng.integers(0, n, size=m)
        if g == 1:
            i[:120] = 7                       # a hub
            i[i == 9] = 8                     # and a node nobody sends to
        if sort_edges:
            order = np.lexsort((j, i))
            i, j = i[order], j[order]
        attrs.append(rng.normal(size=(n, feats)).astype(np.float32))
        wts.append(rng.uniform(0.05, 1.0, size=(m, 1)).astype(np.float32))
        idxs.append(np.stack([i, j], axis=1).astype(np.int64).reshape(m, 2))
        ns.append(ns[-1] + n)
        es.append(es[-1] + m)
    attrs, wts, idxs = np.concatenate(attrs), np.concatenate(wts), np.concatenate(idxs)
    ns, es = np.asarray(ns, np.int64), np.asarray(es, np.int64)
    p = synth.gcn_params(seed=3, depth=2, in_features=feats, units=units, out_units=tuple(mlp_units), random_bias=True)
    model = GCN.make_model(
        inputs=[{"shape": (None, feats), "name": "node_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, 1), "name": "edge_weights", "dtype": "float32", "ragged": True},
                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
        gcn_args={"units": units, "use_bias": True, "activation": "relu", "pooling_method": "sum"},
        depth=2, output_embedding="node", output_to_tensor=False,
        output_mlp={"use_bias": True, "units": mlp_units, "activation": mlp_act})
    model.set_weights(list(p.values()))
    assert model.fused is not None
    ins = [_dev(attrs, ns), _dev(wts, es), _dev(idxs, es)]
    first = model(ins)                       # direct launches
    assert model.fused.last == "eager"
    second = model(ins)                      # captured on this call ...
    third = model(ins)                       # ... a second result buffer (own graph) on this one ...
    del third
    model(ins)                               # ... the ring's last buffer here ...
    before = _ffi.launch_count()
    third = model(ins)                       # ... and from here on a call is a single graph launch
    assert model.fused.last == "graph" and _ffi.launch_count() - before == 1
    layers = model(ins, fused=False).values.cpu().numpy()
    torch.cuda.synchronize()
    got = first.values.cpu().numpy()
    assert np.array_equal(got, second.values.cpu().numpy()) and np.array_equal(got, third.values.cpu().numpy())
    assert second.values.data_ptr() != third.values.data_ptr()

    def oracle(dtype):
        return ko.gcn_forward(ko.to_dtype(p, dtype), ko.R(attrs.astype(dtype), ns), ko.R(wts.astype(dtype), es),
                              ko.R(idxs, es), depth=2, output_mlp_act=tuple(mlp_act)).values
    _check(got, oracle(np.float32), oracle(np.float64))
    _check(layers, oracle(np.float32), oracle(np.float64))


def test_gcn_fused_route_padded_output():
    """``output_to_tensor=True`` (kgcnn/literature/GCN.py:108-109: ChangeTensorType ragged -> tensor) behind the fused
    route: several graphs are padded by the cast layer, a single graph's padded tensor is the value matrix itself."""
    from gcnn_keras_amd.literature import GCN
    for sizes in ([30], [12, 30, 5]):
        rng = np.random.default_rng(len(sizes))
        attrs = rng.normal(size=(sum(sizes), 20)).astype(np.float32)
        ns = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        idx = [np.stack([np.sort(rng.integers(0, n, size=3 * n)), rng.integers(0, n, size=3 * n)], 1) for n in sizes]
        es = np.concatenate([[0], np.cumsum([len(i) for i in idx])]).astype(np.int64)
        idx = np.concatenate(idx).astype(np.int64)
        wts = rng.uniform(0.1, 1.0, size=(len(idx), 1)).astype(np.float32)
        p = synth.gcn_params(seed=5, depth=1, in_features=20, units=32, out_units=(8,), random_bias=True)
        outs = []
        for to_tensor in (False, True):
            model = GCN.make_model(
                inputs=[{"shape": (None, 20), "name": "node_attributes", "dtype": "float32", "ragged": True},
                        {"shape": (None, 1), "name": "edge_weights", "dtype": "float32", "ragged": True},
                        {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
                gcn_args={"units": 32, "use_bias": True, "activation": "relu", "pooling_method": "sum"},
                depth=1, output_embedding="node", output_to_tensor=to_tensor,
                output_mlp={"use_bias": True, "units": [8], "activation": "linear"})
            model.set_weights(list(p.values()))
            ins = [_dev(attrs, ns), _dev(wts, es), _dev(idx, es)]
            model(ins)
            outs.append(model(ins))
            assert model.fused.last == "graph"
        ragged, padded = outs[0].values.cpu().numpy(), outs[1].cpu().numpy()
        assert padded.shape == (len(sizes), max(sizes), 8)
        for g, n in enumerate(sizes):
            assert np.array_equal(padded[g, :n], ragged[ns[g]:ns[g + 1]])
            assert not padded[g, n:].any()


@pytest.mark.parametrize("n,feats,with_edges", [(5, 7, False), (19, 3, True), (16, 16, True), (33, 70, True)])
def test_gcn_fused_route_small_shapes(n, feats, with_edges):
    """Corner shapes of the tile kernel: fewer than 16 nodes, fewer than 16 input features (only the ragged last k block
    exists), an input width that is a multiple of 16 (no ragged block), a batch without any edge."""
    from gcnn_keras_amd.literature import GCN
    rng = np.random.default_rng(n + feats)
    attrs = rng.normal(size=(n, feats)).astype(np.float32)
    ns = np.array([0, n], np.int64)
    if with_edges:
        i = np.sort(rng.integers(0, n, size=3 * n))
        idx = np.stack([i, rng.integers(0, n, size=3 * n)], 1).astype(np.int64)
    else:
        idx = np.zeros((0, 2), np.int64)
    es = np.array([0, len(idx)], np.int64)
    wts = rng.uniform(0.1, 1.0, size=(len(idx), 1)).astype(np.float32)
    p = synth.gcn_params(seed=2, depth=2, in_features=feats, units=64, out_units=(16, 3), random_bias=True)
    model = GCN.make_model(
        inputs=[{"shape": (None, feats), "name": "node_attributes", "dtype": "float32", "ragged": True},
                {"shape": (None, 1), "name": "edge_weights", "dtype": "float32", "ragged": True},
                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
        gcn_args={"units": 64, "use_bias": True, "activation": "relu", "pooling_method": "sum"},
        depth=2, output_embedding="node", output_to_tensor=False,
        output_mlp={"use_bias": True, "units": [16, 3], "activation": ["relu", "softmax"]})
    model.set_weights(list(p.values()))
    ins = [_dev(attrs, ns), _dev(wts, es), _dev(idx, es)]
    first = model(ins).values.cpu().numpy()
    replayed = model(ins).values.cpu().numpy()
    assert model.fused is not None and model.fused.last == "graph" and np.array_equal(first, replayed)

    def oracle(dtype):
        return ko.gcn_forward(ko.to_dtype(p, dtype), ko.R(attrs.astype(dtype), ns), ko.R(wts.astype(dtype), es),
                              ko.R(idx, es), depth=2, output_mlp_act=("relu", "softmax")).values
    _check(first, oracle(np.float32), oracle(np.float64))
