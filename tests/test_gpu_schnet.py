"""SchNet forward through the reference's entry point - ``Schnet.make_model(...)(inputs)`` - on the HIP engine vs the CPU
oracle: the fused route the model takes on its own (8 kernels, direct launch on first sight of a batch, HIP-graph replay
afterwards) and the layer path (``fused=False``).

Tolerance: 1e-5 per ROW (graph) of the output (tests/parity.py; BASELINE.json north_star), with the float64 twin of the
oracle as the error budget: the engine must be as close to float64 truth as the float32 oracle is (within a factor)."""
import os

import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth
from oracle import kgcnn_oracle as ko
from parity import assert_rows_close

pytestmark = pytest.mark.gpu


def _inputs(b):
    from gcnn_keras_amd.ragged import RaggedTensor
    return [RaggedTensor.from_numpy(b["node_number"], b["node_splits"]),
            RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
            RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]


def _oracle(p, b, depth, dtype=np.float32):
    pp = ko.to_dtype(p, dtype)
    return ko.schnet_forward(pp, ko.R(b["node_number"], b["node_splits"]),
                             ko.R(b["node_coordinates"].astype(dtype), b["node_splits"]),
                             ko.R(b["edge_indices"], b["edge_splits"]), depth=depth)


def _model(p, depth=3):
    from gcnn_keras_amd.literature import Schnet
    model = Schnet.make_model(depth=depth)
    model.set_weights(list(p.values()))
    return model


def _shuffled(b, seed=0):
    rng = np.random.default_rng(seed)
    idx = b["edge_indices"].copy()
    for g in range(len(b["edge_splits"]) - 1):
        lo, hi = b["edge_splits"][g], b["edge_splits"][g + 1]
        idx[lo:hi] = idx[lo:hi][rng.permutation(hi - lo)]
    out = dict(b)
    out["edge_indices"] = idx
    return out


@pytest.mark.parametrize("num_graphs,seed", [(6, 11), (1, 5), (128, 1234)])
def test_make_model_takes_the_fused_route(num_graphs, seed, golden_dir):
    """model(inputs) of the reference's builder runs the eight fused kernels: first sight of a batch = one direct launch,
    a re-bound batch = graph replay, eager mode = exactly 8 engine calls; all three give the same bits."""
    from gcnn_keras_amd import _ffi
    b = synth.qm9_like_batch(num_graphs=num_graphs, seed=seed)
    p = synth.schnet_params(seed=7, random_bias=True)
    model = _model(p)
    assert model.fused is not None
    x = _inputs(b)
    out1 = model(x)
    assert model.fused.last == "direct"
    out2 = model(x)
    assert model.fused.last == "graph"
    assert out2.data_ptr() != out1.data_ptr()                 # a model call returns a fresh tensor
    model.fused.mode = "eager"
    before = _ffi.launch_count()
    out3 = model(x)
    assert _ffi.launch_count() - before == 8                  # stage0, 3 x (cfconv + node chain), readout
    model.fused.mode = "auto"
    model.fused.check_flags()
    assert torch.equal(out1, out2) and torch.equal(out1, out3)
    got = out1.cpu().numpy()
    assert got.shape == (num_graphs, 1)
    assert_rows_close(got, _oracle(p, b, 3), _oracle(p, b, 3, np.float64), what="fused make_model")
    if (num_graphs, seed) == (6, 11):
        frozen = np.load(os.path.join(golden_dir, "frozen_schnet_small.npz"))["out"]
        assert_rows_close(got, frozen, what="frozen golden")


@pytest.mark.parametrize("num_graphs,seed", [(6, 11), (1, 5), (128, 1234)])
def test_make_model_layer_path(num_graphs, seed):
    """``fused=False``: the reference's layer sequence op by op (SchNetCFconv and the interaction's node side still use
    their single-layer fused kernels)."""
    from gcnn_keras_amd import _ffi
    b = synth.qm9_like_batch(num_graphs=num_graphs, seed=seed)
    p = synth.schnet_params(seed=7, random_bias=True)
    model = _model(p)
    before = _ffi.launch_count()
    out = model(_inputs(b), fused=False).cpu().numpy()
    assert _ffi.launch_count() - before > 8
    assert out.shape == (num_graphs, 1)
    assert_rows_close(out, _oracle(p, b, 3), _oracle(p, b, 3, np.float64), what="layer path")


def test_make_model_follows_weight_updates_and_in_place_coordinates():
    """``set_weights`` after batches were bound: the kernels read the model's own tensors and the packed images are rebuilt
    (the captured graph stays valid).  New coordinates written into the bound tensor are picked up by the next call."""
    b = synth.qm9_like_batch(num_graphs=9, seed=21)
    p1 = synth.schnet_params(seed=7, random_bias=True)
    p2 = synth.schnet_params(seed=8, random_bias=True)
    model = _model(p1)
    x = _inputs(b)
    model(x), model(x)                                        # bound + captured
    model.set_weights(list(p2.values()))
    got = model(x)
    assert model.fused.last == "graph"
    assert_rows_close(got.cpu().numpy(), _oracle(p2, b, 3), what="after set_weights")
    b2 = dict(b)
    b2["node_coordinates"] = (b["node_coordinates"] * np.float32(0.9)).astype(np.float32)   # same edges, new distances
    x[1].values.copy_(torch.from_numpy(b2["node_coordinates"]).cuda())
    got = model(x)
    assert model.fused.last == "graph"
    assert_rows_close(got.cpu().numpy(), _oracle(p2, b2, 3), what="after in-place coordinate update")


def test_make_model_unsorted_receivers_empty_graphs_and_slot_eviction():
    from gcnn_keras_amd.ragged import RaggedTensor
    p = synth.schnet_params(seed=7, random_bias=True)
    model = _model(p)
    model.fused.max_slots = 2
    # receivers not sorted inside the graphs: the slot takes the stable-sort route (tf.argsort(stable=True), pooling.py:66)
    b = _shuffled(synth.qm9_like_batch(num_graphs=11, seed=8), seed=1)
    x = _inputs(b)
    first, second = model(x), model(x)
    assert not model.fused.slot_of(x).sorted and torch.equal(first, second)
    assert_rows_close(first.cpu().numpy(), _oracle(p, b, 3), what="unsorted receivers")
    # a graph with atoms but no edges, then an empty graph (dropped by PoolingNodes like tf.math.segment_sum does)
    c = synth.qm9_like_batch(num_graphs=5, seed=2)
    c["node_number"] = np.concatenate([c["node_number"], np.array([6., 1., 8.], np.float32)])
    c["node_coordinates"] = np.concatenate([c["node_coordinates"], np.zeros((3, 3), np.float32)])
    n, m = c["node_splits"][-1], c["edge_splits"][-1]
    c["node_splits"] = np.concatenate([c["node_splits"], [n + 3, n + 3]])
    c["edge_splits"] = np.concatenate([c["edge_splits"], [m, m]])
    y = _inputs(c)
    got = model(y).cpu().numpy()
    ref = _oracle(p, c, 3)
    assert got.shape == ref.shape == (6, 1)
    assert_rows_close(got, ref, what="empty graphs")
    # a third batch evicts the least recently used slot; the evicted batch simply binds again
    d = synth.qm9_like_batch(num_graphs=3, seed=30)
    z = _inputs(d)
    model(z)
    assert len(model.fused._slots) == 2 and model.fused.slot_of(x) is None
    again = model(x)
    assert model.fused.last == "direct" and torch.equal(again, first)
    # empty batch of edges (kgcnn's commented-out empty-edge test): every node unconnected
    e = {"node_number": np.array([6., 1.], np.float32), "node_coordinates": np.zeros((2, 3), np.float32),
         "edge_indices": np.zeros((0, 2), np.int64), "node_splits": np.array([0, 2], np.int64),
         "edge_splits": np.array([0, 0], np.int64)}
    got = model(_inputs(e)).cpu().numpy()
    assert_rows_close(got, _oracle(p, e, 3), what="no edges")
    model.fused.check_flags()


def test_make_model_routes_to_layers_when_it_must():
    """Configurations / inputs outside the fused kernels run the layer path: other widths, gradients requested, integer
    node numbers; ``fused=True`` then raises."""
    from gcnn_keras_amd.literature import Schnet
    b = synth.qm9_like_batch(num_graphs=4, seed=3)
    small = Schnet.make_model(depth=2, interaction_args={"units": 64}, last_mlp={"units": [64, 32]},
                              output_mlp={"units": [32, 1]})
    assert small.fused is None
    assert small(_inputs(b)).shape == (4, 1)
    p = synth.schnet_params(seed=7, random_bias=True)
    model = _model(p)
    x = _inputs(b)
    x[1].values.requires_grad_(True)                          # forces need the reverse pass: layer path
    with torch.enable_grad():
        e = model(x)
        assert e.requires_grad
        with pytest.raises(ValueError):
            model(x, fused=True)
    assert_rows_close(e.detach().cpu().numpy(), _oracle(p, b, 3), what="grad route")


def test_schnet_config_surface():
    from gcnn_keras_amd.literature import Schnet
    with pytest.raises(ValueError):
        Schnet.make_model(not_a_kwarg=1)                 # update_model_kwargs rejects unknown keys
    m = Schnet.make_model()                              # model_default depth is 4 (kgcnn/literature/Schnet.py:34)
    names = [n for n, _ in m.weights]
    assert len(names) == 1 + 2 + 4 * 9 + 4 + 4
    shapes = [tuple(t.shape) for _, t in m.weights]
    assert shapes[0] == (95, 64) and shapes[1] == (64, 128)
    assert shapes[3] == (20, 128) and shapes[7] == (128, 128)   # cfconv.dense1 kernel, interaction.dense1 (no bias)
    cfg = m.layers[2].get_config()
    assert cfg["units"] == 128 and cfg["cfconv_pool"] == "sum" and cfg["activation"] == "kgcnn>shifted_softplus"
    assert m.fused is not None                           # the reference's defaults fit the fused kernels at any depth
