"""Set2Set readout, NMPN message / update layers and the MEGNet / NMPN model builders on the HIP
engine vs the CPU oracle.  These results are 'parity unpinned' w.r.t. the reference (it holds no value fixture for
them); the recurrent cells follow Keras' documented LSTM / GRU arithmetic, restated in oracle/kgcnn_oracle.py."""
import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth
from helpers import dev
from oracle import kgcnn_oracle as ko
from parity import assert_rows_close

pytestmark = pytest.mark.gpu


def _randomise(obj, seed):
    """Random values for every weight (zero-initialised biases included), returned as the get_weights() list."""
    rng = np.random.default_rng(seed)
    new = []
    for _, t in obj.weights:
        shape = tuple(t.shape)
        scale = 0.3 if len(shape) == 1 or shape[0] == 2 else min(1.0, 1.5 / np.sqrt(shape[0]))
        new.append(rng.normal(scale=scale, size=shape).astype(np.float32))
    obj.set_weights(new)
    return new


@pytest.mark.parametrize("pooling_method", ["sum", "mean"])
@pytest.mark.parametrize("init_qstar", ["0", "mean"])
def test_pooling_set2set_vs_oracle(pooling_method, init_qstar):
    from gcnn_keras_amd.layers.pool.set2set import PoolingSet2Set
    rng = np.random.default_rng(2)
    lens = [5, 1, 0, 9, 3]                                  # incl. an empty set: q from the LSTM, r = 0
    splits = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    vals = rng.normal(size=(splits[-1], 16)).astype(np.float32)
    lay = PoolingSet2Set(16, T=3, pooling_method=pooling_method, init_qstar=init_qstar)
    lay.ensure_built((None, None, 16))
    w = _randomise(lay, 5)
    got = lay(dev(vals, splits)).cpu().numpy()
    ref = ko.pooling_set2set(ko.R(vals, splits), w[0], w[2], T=3, pooling_method=pooling_method, init_qstar=init_qstar)
    ref64 = ko.pooling_set2set(ko.R(vals.astype(np.float64), splits), w[0].astype(np.float64), w[2].astype(np.float64),
                               T=3, pooling_method=pooling_method, init_qstar=init_qstar)
    assert got.shape == ref.shape == (5, 1, 32)
    assert_rows_close(got[:, 0], ref[:, 0], ref64[:, 0], what="Set2Set")
    cfg = lay.get_config()
    assert cfg["channels"] == 16 and cfg["T"] == 3 and cfg["init_qstar"] == init_qstar and cfg["unit_forget_bias"] is True
    with pytest.raises(TypeError):
        PoolingSet2Set(16, pooling_method="max")


def test_nmpn_layers_vs_oracle():
    from gcnn_keras_amd.layers.conv.mpnn_conv import GRUUpdate, MatMulMessages, TrafoEdgeNetMessages
    b = synth.qm9_like_batch(num_graphs=4, seed=12)
    rng = np.random.default_rng(1)
    n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
    nodes = rng.normal(size=(n, 64)).astype(np.float32)
    upd = rng.normal(size=(n, 128)).astype(np.float32)
    edges = rng.normal(size=(m, 24)).astype(np.float32)
    gru = GRUUpdate(64)
    gru.ensure_built([(None, None, 64), (None, None, 128)])
    wg = _randomise(gru, 3)
    got = gru([dev(nodes, b["node_splits"]), dev(upd, b["node_splits"])]).values.cpu().numpy()
    ref = ko.gru_update(ko.R(nodes, b["node_splits"]), ko.R(upd, b["node_splits"]), *wg).values
    assert_rows_close(got, ref, what="GRUUpdate")
    assert gru.get_config()["reset_after"] is True and gru.get_config()["units"] == 64
    trafo = TrafoEdgeNetMessages(target_shape=(64, 64))
    trafo.ensure_built((None, None, 24))
    wt = _randomise(trafo, 4)
    mats = trafo(dev(edges, b["edge_splits"]))
    assert tuple(mats.values.shape) == (m, 64, 64)
    ref_m = ko.trafo_edge_net_messages(ko.R(edges, b["edge_splits"]), wt[0], wt[1], (64, 64))
    assert_rows_close(mats.values.cpu().numpy(), ref_m.values, what="TrafoEdgeNetMessages")
    msg = rng.normal(size=(m, 64)).astype(np.float32)
    got = MatMulMessages()([mats, dev(msg, b["edge_splits"])]).values.cpu().numpy()
    ref = ko.matmul_messages(ko.R(mats.values.cpu().numpy(), b["edge_splits"]), ko.R(msg, b["edge_splits"])).values
    assert_rows_close(got, ref, what="MatMulMessages")
    # odd sizes take the generic matvec kernel
    odd = torch.from_numpy(rng.normal(size=(7, 5, 9)).astype(np.float32)).cuda()
    vec = torch.from_numpy(rng.normal(size=(7, 9)).astype(np.float32)).cuda()
    from gcnn_keras_amd.ragged import RaggedTensor
    sp = torch.tensor([0, 7], device="cuda")
    got = MatMulMessages()([RaggedTensor(odd, sp), RaggedTensor(vec, sp)]).values.cpu().numpy()
    assert np.allclose(got, np.einsum("mrc,mc->mr", odd.cpu().numpy(), vec.cpu().numpy()), rtol=1e-5, atol=1e-6)


def test_megnet_make_model_vs_oracle():
    from gcnn_keras_amd.literature import Megnet
    b = synth.qm9_like_batch(num_graphs=7, seed=41)
    env = np.array([3, 0, 7, 1, 99, 5, 2], np.float32)           # graph_attributes through an Embedding (Megnet.py:123)
    model = Megnet.make_model()
    w = _randomise(model, 9)
    assert len(w) == 108
    inputs = [dev(b["node_number"], b["node_splits"]), dev(b["node_coordinates"], b["node_splits"]),
              dev(b["edge_indices"], b["edge_splits"]), torch.from_numpy(env).cuda()]
    out = model(inputs)
    again = model(inputs)                                        # auto_graph: the second call replays one HIP graph
    assert model.last_route == "graph" and torch.equal(out, again)
    args = (ko.R(b["node_number"], b["node_splits"]), ko.R(b["node_coordinates"], b["node_splits"]),
            ko.R(b["edge_indices"], b["edge_splits"]), env)
    ref = ko.megnet_forward(w, *args)
    w64 = [x.astype(np.float64) for x in w]
    ref64 = ko.megnet_forward(w64, args[0], ko.R(b["node_coordinates"].astype(np.float64), b["node_splits"]), args[2],
                              env)
    got = out.cpu().numpy()
    assert got.shape == ref.shape == (7, 1)
    assert_rows_close(got, ref, ref64, what="Megnet.make_model")
    with pytest.raises(ValueError):
        Megnet.make_model(output_embedding="node")
    plain = Megnet.make_model(use_set2set=False, nblocks=1)
    assert plain(inputs).shape == (7, 1)


def test_nmpn_make_model_vs_oracle():
    from gcnn_keras_amd.literature import NMPN
    b = synth.qm9_like_batch(num_graphs=5, seed=43)
    rng = np.random.default_rng(7)
    edge_number = rng.integers(0, 5, size=int(b["edge_splits"][-1])).astype(np.float32)
    model = NMPN.make_model()
    w = _randomise(model, 10)
    inputs = [dev(b["node_number"], b["node_splits"]), dev(edge_number, b["edge_splits"]),
              dev(b["edge_indices"], b["edge_splits"])]
    got = model(inputs).cpu().numpy()
    args = (ko.R(b["node_number"], b["node_splits"]), ko.R(edge_number, b["edge_splits"]),
            ko.R(b["edge_indices"], b["edge_splits"]))
    ref = ko.nmpn_forward(w, *args)
    ref64 = ko.nmpn_forward([x.astype(np.float64) for x in w], *args)
    assert got.shape == ref.shape == (5, 1)
    assert_rows_close(got, ref, ref64, what="NMPN.make_model")
    node_out = NMPN.make_model(output_embedding="node", depth=1)
    assert node_out(inputs).shape[0] == 5


def test_gather_embedding_general_route():
    """kgcnn/layers/gather.py:121-138: concat / split along an axis other than the index axis (node features with two
    dense axes)."""
    from gcnn_keras_amd.layers.gather import GatherEmbedding
    b = synth.qm9_like_batch(num_graphs=3, seed=2)
    rng = np.random.default_rng(0)
    n = int(b["node_splits"][-1])
    x = rng.normal(size=(n, 3, 4)).astype(np.float32)             # (batch, [N], 3, 4)
    nodes, idx = dev(x, b["node_splits"]), dev(b["edge_indices"], b["edge_splits"])
    sh = ko._shift(ko.R(x, b["node_splits"]), ko.R(b["edge_indices"], b["edge_splits"]))
    full = x[sh]                                                  # (M, 2, 3, 4) = tf.gather(batch_dims=1)
    got = GatherEmbedding(concat_axis=3)([nodes, idx]).values.cpu().numpy()
    want = np.concatenate([full[:, :, i] for i in range(3)], axis=2)      # pick i along axis 3, concat along axis 3
    assert np.array_equal(got, want)
    parts = GatherEmbedding(concat_axis=None, split_axis=3, split_indices=[2, 0])([nodes, idx])
    assert np.array_equal(parts[0].values.cpu().numpy(), full[:, :, 2])
    assert np.array_equal(parts[1].values.cpu().numpy(), full[:, :, 0])
    with pytest.raises(NotImplementedError):
        GatherEmbedding(axis=2)([nodes, idx])
