"""GPU parity tests: every layer of the hot path, through the C ABI, against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): bit-exact for index / gather work; float segment reductions within 1e-5 relative
(the sequential CSR kernel is in fact compared for equality where the accumulation order is defined); dense /
transcendental layers within 1e-5 of the output scale.
"""
import os

import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth
from oracle import kgcnn_oracle as ko

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # relative to the output scale, the tolerance north_star states for float work


def _dev(r):
    from gcnn_keras_amd.ragged import RaggedTensor
    return RaggedTensor.from_numpy(r.values, r.row_splits)


def _close(got, ref, rtol=RTOL):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = max(float(np.max(np.abs(ref))) if ref.size else 0.0, 1e-30)
    err = float(np.max(np.abs(got - ref))) if ref.size else 0.0
    assert err <= rtol * scale, "max abs err %g vs scale %g" % (err, scale)


def _exact(got, ref):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(got, ref)


def _rand_case(seed, n_graphs=7, f=16, sort=False, max_nodes=9, max_edges=40):
    rng = np.random.default_rng(seed)
    n_len = rng.integers(0, max_nodes, size=n_graphs)
    n_len[rng.integers(n_graphs)] = max_nodes
    e_len = np.array([rng.integers(0, max_edges) if n > 0 else 0 for n in n_len])
    idx = np.concatenate([rng.integers(0, max(n, 1), size=(m, 2)) for n, m in zip(n_len, e_len)]).astype(np.int64)
    if sort:
        parts, o = [], 0
        for m in e_len:
            blk = idx[o:o + m]
            parts.append(blk[np.lexsort((blk[:, 1], blk[:, 0]))])
            o += m
        idx = np.concatenate(parts)
    nodes = ko.ragged_from_row_lengths(rng.normal(size=(n_len.sum(), f)).astype(np.float32), n_len)
    edges = ko.ragged_from_row_lengths(rng.normal(size=(e_len.sum(), f)).astype(np.float32), e_len)
    w = ko.ragged_from_row_lengths(rng.uniform(0.1, 1, size=(e_len.sum(), 1)).astype(np.float32), e_len)
    return nodes, edges, ko.ragged_from_row_lengths(idx, e_len), w


# ------------------------------------------------------------------------------------------------- index / gather
def test_partition_row_indexing_bit_exact():
    from gcnn_keras_amd.ops.partition import partition_row_indexing
    nodes, _, idx, _ = _rand_case(0)
    out = partition_row_indexing(torch.from_numpy(idx.values).cuda(), torch.from_numpy(nodes.row_splits).cuda(),
                                 torch.from_numpy(ko.row_lengths(idx)).cuda(), "row_splits", "row_length")
    ref = ko._shift(nodes, idx)
    assert out.dtype == torch.int64
    _exact(out, ref)
    back = partition_row_indexing(out, torch.from_numpy(nodes.row_splits).cuda(),
                                  torch.from_numpy(ko.row_lengths(idx)).cuda(), "row_splits", "row_length",
                                  from_indexing="batch", to_indexing="sample")
    _exact(back, idx.values)
    # docstring example of kgcnn/ops/partition.py:112-120
    out = partition_row_indexing(torch.tensor([0, 0, 1, 1]).cuda(), torch.tensor([2, 2]).cuda(),
                                 torch.tensor([3, 1]).cuda(), "row_lengths", "row_lengths")
    assert out.cpu().tolist() == [0, 0, 1, 3]


def test_gather_nodes_reference_case_bit_exact(golden_dir):
    from gcnn_keras_amd.layers.gather import GatherNodes
    d = np.load(os.path.join(golden_dir, "gather_case.npz"))
    node = ko.ragged_from_row_lengths(np.concatenate([d["n0"], d["n1"]]), [8, 15])
    idx = ko.ragged_from_row_lengths(np.concatenate([d["ei0"], d["ei1"]]), [14, 28])
    g = GatherNodes()([_dev(node), _dev(idx)])
    assert g.shape == (2, None, 2)
    _exact(g[1], np.reshape(d["n1"][d["ei1"]], (28, 2)))          # test/test_gather.py:28-35
    g2 = GatherNodes(concat_axis=None)([_dev(node), _dev(idx)])
    _exact(g2[1], d["n1"][d["ei1"]])                               # test/test_gather.py:37-44


@pytest.mark.parametrize("f", [1, 3, 16, 128])
def test_gather_layers_bit_exact(f):
    from gcnn_keras_amd.layers.gather import (GatherNodes, GatherNodesIngoing, GatherNodesOutgoing,
                                              GatherNodesSelection, GatherState)
    nodes, _, idx, _ = _rand_case(1, f=f)
    dn, di = _dev(nodes), _dev(idx)
    _exact(GatherNodes()([dn, di]).values, ko.gather_nodes(nodes, idx).values)
    sp = GatherNodes(concat_axis=None, split_axis=2)([dn, di])
    for a, b in zip(sp, ko.gather_nodes(nodes, idx, concat_axis=None, split_axis=2)):
        _exact(a.values, b.values)
    _exact(GatherNodesOutgoing()([dn, di]).values, ko.gather_nodes_outgoing(nodes, idx).values)
    _exact(GatherNodesIngoing()([dn, di]).values, ko.gather_nodes_ingoing(nodes, idx).values)
    sel = GatherNodesSelection([1, 0])([dn, di])
    _exact(sel[0].values, ko.gather_nodes_selection(nodes, idx, [1, 0])[0].values)
    state = np.random.default_rng(3).normal(size=(nodes.row_splits.shape[0] - 1, 5)).astype(np.float32)
    _exact(GatherState()([torch.from_numpy(state).cuda(), dn]).values, ko.gather_state(state, nodes).values)
    with pytest.raises(ValueError):
        GatherNodes(concat_axis=2, split_axis=2)


def test_gather_rank3_values_painn_shape():
    from gcnn_keras_amd.layers.gather import GatherNodesOutgoing
    nodes, _, idx, _ = _rand_case(2, f=8)
    v = np.random.default_rng(4).normal(size=(nodes.values.shape[0], 3, 8)).astype(np.float32)
    rv = ko.R(v, nodes.row_splits)
    _exact(GatherNodesOutgoing()([_dev(rv), _dev(idx)]).values, ko.gather_nodes_outgoing(rv, idx).values)


def test_empty_edges_and_empty_batch():
    from gcnn_keras_amd.layers.gather import GatherNodes
    from gcnn_keras_amd.layers.pooling import PoolingLocalEdges, PoolingNodes
    nodes = ko.ragged_from_row_lengths(np.arange(12, dtype=np.float32).reshape(6, 2), [2, 4])
    idx = ko.ragged_from_row_lengths(np.zeros((0, 2), np.int64), [0, 0])       # the reference's commented-out case
    edges = ko.ragged_from_row_lengths(np.zeros((0, 2), np.float32), [0, 0])
    g = GatherNodes()([_dev(nodes), _dev(idx)])
    assert tuple(g.values.shape) == (0, 4)
    out = PoolingLocalEdges("sum")([_dev(nodes), _dev(edges), _dev(idx)])
    _exact(out.values, np.zeros((6, 2), np.float32))
    pooled = PoolingNodes("sum")(_dev(nodes))
    _exact(pooled, ko.pooling_nodes(nodes, "sum"))


def test_out_of_range_index_is_flagged_not_faulting():
    from gcnn_keras_amd.layers.gather import GatherNodes
    nodes = ko.ragged_from_row_lengths(np.ones((5, 4), np.float32), [2, 3])
    idx = ko.ragged_from_row_lengths(np.array([[0, 1], [1, 7], [2, 0]], np.int64), [2, 1])
    GatherNodes()([_dev(nodes), _dev(idx)])  # must not fault
    with pytest.raises(IndexError):
        GatherNodes(ragged_validate=True)([_dev(nodes), _dev(idx)])


# ------------------------------------------------------------------------------------------------- pooling
@pytest.mark.parametrize("method", ["sum", "mean", "max", "min", "segment_sum", "reduce_mean"])
@pytest.mark.parametrize("sort", [False, True])
@pytest.mark.parametrize("f", [1, 16, 130])
def test_pooling_local_edges(method, sort, f):
    from gcnn_keras_amd.layers.pooling import PoolingLocalEdges
    nodes, edges, idx, _ = _rand_case(5, f=f, sort=sort)
    out = PoolingLocalEdges(pooling_method=method)([_dev(nodes), _dev(edges), _dev(idx)])
    ref = ko.pooling_local_edges(nodes, edges, idx, method)
    _exact(out.row_splits, nodes.row_splits)
    if "mean" in method:
        _close(out.values, ref.values, rtol=1e-6)
    else:
        _exact(out.values, ref.values)  # same sequential accumulation order as the oracle
    if sort:
        out2 = PoolingLocalEdges(pooling_method=method, is_sorted=True)([_dev(nodes), _dev(edges), _dev(idx)])
        _exact(out2.values, out.values.cpu().numpy())


def test_pooling_defaults_and_errors():
    from gcnn_keras_amd.layers.pooling import PoolingLocalEdges, PoolingNodes, PoolingLocalMessages
    assert PoolingLocalEdges().pooling_method == "mean"     # kgcnn/layers/pooling.py:27
    assert PoolingNodes().pooling_method == "mean"          # kgcnn/layers/pooling.py:194
    assert PoolingLocalMessages is PoolingLocalEdges
    nodes, edges, idx, _ = _rand_case(6)
    with pytest.raises(TypeError):
        PoolingLocalEdges(pooling_method="median")([_dev(nodes), _dev(edges), _dev(idx)])
    cfg = PoolingLocalEdges(pooling_method="sum", is_sorted=True).get_config()
    assert cfg["pooling_method"] == "sum" and cfg["pooling_index"] == 0 and cfg["is_sorted"] is True
    assert cfg["has_unconnected"] is True and cfg["node_indexing"] == "sample"


def test_pooling_has_unconnected_false_rows():
    from gcnn_keras_amd.layers.pooling import PoolingLocalEdges
    nodes = ko.ragged_from_row_lengths(np.ones((6, 3), np.float32), [3, 3])
    idx = ko.ragged_from_row_lengths(np.array([[0, 1], [1, 0], [0, 2], [1, 0]], np.int64), [2, 2])
    edges = ko.ragged_from_row_lengths(np.arange(12, dtype=np.float32).reshape(4, 3), [2, 2])
    out = PoolingLocalEdges("sum", has_unconnected=False)([_dev(nodes), _dev(edges), _dev(idx)])
    ref = ko.pooling_local_edges(nodes, edges, idx, "sum", has_unconnected=False)
    _exact(out.values, ref.values)  # max(receiver)+1 = 5 rows, not 6 (SURVEY 8a note 5)
    assert out.values.shape[0] == 5


@pytest.mark.parametrize("method", ["sum", "mean", "max"])
@pytest.mark.parametrize("normalize", [False, True])
def test_pooling_weighted_local_edges(method, normalize):
    from gcnn_keras_amd.layers.pooling import PoolingWeightedLocalEdges
    nodes, edges, idx, w = _rand_case(7)
    out = PoolingWeightedLocalEdges(pooling_method=method, normalize_by_weights=normalize)(
        [_dev(nodes), _dev(edges), _dev(idx), _dev(w)])
    ref = ko.pooling_weighted_local_edges(nodes, edges, idx, w, method, normalize_by_weights=normalize)
    _close(out.values, ref.values, rtol=2e-6)


@pytest.mark.parametrize("method", ["sum", "mean", "max", "min"])
def test_pooling_nodes(method):
    from gcnn_keras_amd.layers.pooling import PoolingNodes, PoolingWeightedNodes
    nodes, _, _, _ = _rand_case(8, f=64)
    out = PoolingNodes(pooling_method=method)(_dev(nodes))
    ref = ko.pooling_nodes(nodes, method)
    if method == "mean":
        _close(out, ref, rtol=1e-6)
    else:
        _exact(out, ref)
    w = ko.R(np.random.default_rng(1).uniform(0.5, 1, size=(nodes.values.shape[0], 1)).astype(np.float32),
             nodes.row_splits)
    _close(PoolingWeightedNodes(pooling_method=method)([_dev(nodes), _dev(w)]),
           ko.pooling_weighted_nodes(nodes, w, method), rtol=2e-6)


def test_pooling_nodes_trailing_empty_graphs_dropped():
    from gcnn_keras_amd.layers.pooling import PoolingNodes
    r = ko.R(np.arange(12, dtype=np.float32).reshape(6, 2), np.array([0, 2, 2, 6, 6, 6], dtype=np.int64))
    out = PoolingNodes("sum")(_dev(r))
    _exact(out, ko.pooling_nodes(r, "sum"))
    assert out.shape[0] == 3


def test_attention_pooling_known_answer_and_random():
    from gcnn_keras_amd.layers.pooling import PoolingLocalEdgesAttention, PoolingNodesAttention
    nodes = ko.ragged_from_row_lengths(np.array([[1.0], [1.0]], np.float32), [1, 1])
    edges = ko.ragged_from_row_lengths(np.array([[100.0], [0.0], [100.0], [0.0]], np.float32), [2, 2])
    att = ko.ragged_from_row_lengths(np.array([[0.0], [1.0], [0.0], [1.0]], np.float32), [2, 2])
    idx = ko.ragged_from_row_lengths(np.zeros((4, 2), dtype=np.int64), [2, 2])
    res = PoolingLocalEdgesAttention()([_dev(nodes), _dev(edges), _dev(att), _dev(idx)])
    assert abs(float(res[0][0, 0]) - 100.0 / (np.exp(1) + 1)) < 1e-4      # test/test_conv_attention.py:34-43
    nodes, edges, idx, w = _rand_case(9)
    out = PoolingLocalEdgesAttention()([_dev(nodes), _dev(edges), _dev(w), _dev(idx)])
    _close(out.values, ko.pooling_local_edges_attention(nodes, edges, w, idx).values, rtol=2e-6)
    a = ko.R(np.random.default_rng(2).normal(size=(nodes.values.shape[0], 1)).astype(np.float32), nodes.row_splits)
    _close(PoolingNodesAttention()([_dev(nodes), _dev(a)]), ko.pooling_nodes_attention(nodes, a), rtol=2e-6)


@pytest.mark.parametrize("method", ["sum", "max", "min"])
def test_relational_pooling(method):
    from gcnn_keras_amd.layers.pooling import RelationalPoolingLocalEdges
    nodes, edges, idx, _ = _rand_case(10, f=4)
    rel = ko.R(np.random.default_rng(3).integers(0, 3, size=edges.values.shape[0]).astype(np.int64), idx.row_splits)
    out = RelationalPoolingLocalEdges(num_relations=3, pooling_method=method)(
        [_dev(nodes), _dev(edges), _dev(idx), _dev(rel)])
    ref = ko.relational_pooling_local_edges(nodes, edges, idx, rel, 3, method)
    _close(out.values, ref.values, rtol=2e-6)   # atomic order is free for sum; max/min are exact


# ------------------------------------------------------------------------------------------------- dense / modules
@pytest.mark.parametrize("shape", [(37, 20, 128), (130, 128, 128), (65, 64, 1), (9, 1433, 64), (200, 128, 384)])
@pytest.mark.parametrize("act", ["linear", "kgcnn>shifted_softplus", "relu", "swish"])
def test_dense(shape, act):
    from gcnn_keras_amd.layers.modules import Dense
    r, k, u = shape
    rng = np.random.default_rng(r + k)
    x = rng.normal(size=(r, k)).astype(np.float32)
    w = synth.glorot_uniform(rng, k, u)
    b = rng.uniform(-0.1, 0.1, size=u).astype(np.float32)
    lay = Dense(u, activation=act)
    lay.ensure_built((None, None, k))
    lay.set_weights([w, b])
    rag = ko.ragged_from_row_lengths(x, [r // 2, r - r // 2])
    out = lay(_dev(rag))
    ref = ko.dense(ko.to_dtype(rag, np.float64), w.astype(np.float64), b.astype(np.float64), act)
    _close(out.values, ref.values.astype(np.float32), rtol=1e-5)
    cfg = lay.get_config()
    assert cfg["units"] == u and cfg["activation"] == act and cfg["use_bias"] is True


def test_dense_rank3_and_no_bias():
    from gcnn_keras_amd.layers.modules import Dense
    rng = np.random.default_rng(0)
    x = rng.normal(size=(11, 3, 32)).astype(np.float32)
    w = synth.glorot_uniform(rng, 32, 48)
    lay = Dense(48, use_bias=False)
    lay.ensure_built((None, None, 3, 32))
    lay.set_weights([w])
    out = lay(_dev(ko.ragged_from_row_lengths(x, [5, 6])))
    _close(out.values, np.matmul(x.astype(np.float64), w.astype(np.float64)).astype(np.float32))


def test_lazy_layers_and_broadcast():
    from gcnn_keras_amd.layers.modules import (ExpandDims, LazyAdd, LazyAverage, LazyConcatenate, LazyMultiply,
                                                LazySubtract)
    rng = np.random.default_rng(1)
    a = ko.ragged_from_row_lengths(rng.normal(size=(9, 6)).astype(np.float32), [4, 5])
    b = ko.ragged_from_row_lengths(rng.normal(size=(9, 6)).astype(np.float32), [4, 5])
    c = ko.ragged_from_row_lengths(rng.normal(size=(9, 1)).astype(np.float32), [4, 5])
    _exact(LazyAdd()([_dev(a), _dev(b)]).values, a.values + b.values)
    _exact(LazySubtract()([_dev(a), _dev(b)]).values, a.values - b.values)
    _exact(LazyMultiply()([_dev(a), _dev(c)]).values, a.values * c.values)
    _close(LazyAverage()([_dev(a), _dev(b)]).values, (a.values + b.values) * np.float32(0.5), rtol=1e-6)
    _exact(LazyConcatenate(axis=-1)([_dev(a), _dev(c), _dev(b)]).values,
           np.concatenate([a.values, c.values, b.values], axis=-1))
    # PaiNN broadcasts: (M,1,F)*(M,3,F) and (M,1,F)*(M,3,1)  (kgcnn/layers/conv/painn_conv.py:108-112)
    sw = ko.ragged_from_row_lengths(rng.normal(size=(9, 8)).astype(np.float32), [4, 5])
    vj = ko.ragged_from_row_lengths(rng.normal(size=(9, 3, 8)).astype(np.float32), [4, 5])
    rij = ko.ragged_from_row_lengths(rng.normal(size=(9, 3)).astype(np.float32), [4, 5])
    e1 = ExpandDims(axis=-2)(_dev(sw))
    assert tuple(e1.values.shape) == (9, 1, 8)
    _exact(LazyMultiply()([e1, _dev(vj)]).values, sw.values[:, None, :] * vj.values)
    e2 = ExpandDims(axis=-1)(_dev(rij))
    assert tuple(e2.values.shape) == (9, 3, 1)
    _exact(LazyMultiply()([e1, e2]).values, sw.values[:, None, :] * rij.values[:, :, None])


def test_embedding_and_mlp():
    from gcnn_keras_amd.layers.mlp import GraphMLP
    from gcnn_keras_amd.layers.modules import OptionalInputEmbedding
    rng = np.random.default_rng(2)
    z = ko.ragged_from_row_lengths(rng.choice([1., 6., 7., 8., 9.], size=13).astype(np.float32), [6, 7])
    emb = OptionalInputEmbedding(input_dim=95, output_dim=64, use_embedding=True)
    emb.ensure_built((None, None))
    table = emb.get_weights()[0]
    assert table.shape == (95, 64) and np.abs(table).max() <= 0.05
    _exact(emb(_dev(z)).values, ko.embedding(z, table).values)
    mlp = GraphMLP(units=[32, 8], activation=["kgcnn>shifted_softplus", "linear"], use_bias=[True, False])
    x = ko.ragged_from_row_lengths(rng.normal(size=(13, 16)).astype(np.float32), [6, 7])
    mlp.ensure_built((None, None, 16))
    ws = mlp.get_weights()
    assert [w.shape for w in ws] == [(16, 32), (32,), (32, 8)]
    ref = ko.mlp(x, [(ws[0], ws[1], "kgcnn>shifted_softplus"), (ws[2], None, "linear")])
    _close(mlp(_dev(x)).values, ref.values)


# ------------------------------------------------------------------------------------------------- geometry
def test_bessel_basis_reference_asset(golden_dir):
    from gcnn_keras_amd.layers.geom import BesselBasisLayer, NodeDistanceEuclidean, NodePosition
    d = np.load(os.path.join(golden_dir, "bessel_basis_reference.npz"))
    x = ko.ragged_from_row_lengths(np.concatenate([d["x0"], d["x1"]]).astype(np.float32), [5, 11])
    ei = ko.ragged_from_row_lengths(np.concatenate([d["ei0"], d["ei1"]]), [20, 108])
    a, b = NodePosition()([_dev(x), _dev(ei)])
    dist = NodeDistanceEuclidean()([a, b])
    bes = BesselBasisLayer(10, 5.0)(dist)
    assert np.max(np.abs(d["bessel_basis_0"] - bes[0].cpu().numpy())) < 1e-5     # test/test_geom.py:127
    assert np.max(np.abs(d["bessel_basis_1"] - bes[1].cpu().numpy())) < 1e-5     # test/test_geom.py:128


def test_geometry_layers():
    from gcnn_keras_amd.layers.geom import (CosCutOffEnvelope, EdgeDirectionNormalized, EuclideanNorm,
                                            GaussBasisLayer, NodeDistanceEuclidean, NodePosition, ScalarProduct)
    b = synth.qm9_like_batch(num_graphs=5, seed=3)
    xyz = ko.R(b["node_coordinates"], b["node_splits"])
    idx = ko.R(b["edge_indices"], b["edge_splits"])
    p1, p2 = NodePosition()([_dev(xyz), _dev(idx)])
    o1, o2 = ko.node_position(xyz, idx)
    _exact(p1.values, o1.values)
    _exact(p2.values, o2.values)
    dist = NodeDistanceEuclidean()([p1, p2])
    od = ko.node_distance_euclidean(o1, o2)
    _close(dist.values, od.values, rtol=1e-6)
    _close(EdgeDirectionNormalized()([p1, p2]).values, ko.edge_direction_normalized(o1, o2).values, rtol=1e-6)
    _close(GaussBasisLayer(bins=20, distance=4, sigma=0.4)(dist).values, ko.gauss_basis(od, 20, 4.0, 0.4).values)
    _close(CosCutOffEnvelope(5.0)(dist).values, ko.cos_cutoff_envelope(od, 5.0).values)
    _close(CosCutOffEnvelope(None)(dist).values, ko.cos_cutoff_envelope(od, None).values)
    v = ko.R(np.random.default_rng(0).normal(size=(xyz.values.shape[0], 3, 8)).astype(np.float32), xyz.row_splits)
    _close(EuclideanNorm(axis=2)(_dev(v)).values, ko.euclidean_norm(v, axis=2).values, rtol=1e-6)
    _close(ScalarProduct(axis=2)([_dev(v), _dev(v)]).values, ko.scalar_product(v, v, axis=2).values, rtol=1e-6)
    # zero distance: divide_no_nan gives 0 direction, sqrt(0) = 0 distance
    z = ko.ragged_from_row_lengths(np.zeros((2, 3), np.float32), [2])
    e = ko.ragged_from_row_lengths(np.array([[0, 1]], np.int64), [1])
    q1, q2 = NodePosition()([_dev(z), _dev(e)])
    assert float(NodeDistanceEuclidean()([q1, q2]).values.abs().max()) == 0.0
    assert float(EdgeDirectionNormalized()([q1, q2]).values.abs().max()) == 0.0


def test_change_tensor_type_padded_mask():
    from gcnn_keras_amd.layers.casting import ChangeTensorType
    r = ko.ragged_from_row_lengths(np.arange(21, dtype=np.float32).reshape(7, 3), [2, 0, 5])
    padded, mask = ChangeTensorType(input_tensor_type="ragged", output_tensor_type="mask")(_dev(r))
    rp, rm = ko.ragged_to_padded(r)
    _exact(padded, rp)
    _exact(mask, rm)


# ------------------------------------------------------------------------------------------------- toy README model
def test_readme_toy_model_config1():
    """BASELINE config 1: GatherNodes -> Dense(10, relu) -> PoolingLocalMessages(mean) -> concat -> Dense(1) ->
    PoolingNodes(mean) (reference README.md:106-111)."""
    from gcnn_keras_amd.layers.gather import GatherNodes
    from gcnn_keras_amd.layers.modules import Dense, LazyConcatenate
    from gcnn_keras_amd.layers.pooling import PoolingLocalMessages, PoolingNodes
    t = synth.toy_batch()
    n = ko.R(t["node_attributes"], t["node_splits"])
    ei = ko.R(t["edge_indices"], t["edge_splits"])
    rng = np.random.default_rng(0)
    w1, w2 = synth.glorot_uniform(rng, 6, 10), synth.glorot_uniform(rng, 13, 1)
    dn, dei = _dev(n), _dev(ei)
    g = GatherNodes()([dn, dei])
    _exact(g.values, ko.gather_nodes(n, ei).values)
    d1 = Dense(10, activation="relu"); d1.ensure_built((None, None, 6)); d1.set_weights([w1, np.zeros(10, np.float32)])
    d2 = Dense(1); d2.ensure_built((None, None, 13)); d2.set_weights([w2, np.zeros(1, np.float32)])
    msg = d1(g)
    pooled = PoolingLocalMessages()([dn, msg, dei])
    cat = LazyConcatenate(axis=-1)([dn, pooled])
    out = PoolingNodes()(d2(cat))
    # oracle
    om = ko.dense(ko.gather_nodes(n, ei), w1, None, "relu")
    op = ko.pooling_local_edges(n, om, ei, "mean")
    oo = ko.pooling_nodes(ko.dense(ko.lazy_concatenate([n, op]), w2, None, None), "mean")
    _close(pooled.values, op.values, rtol=1e-6)
    _close(out, oo, rtol=1e-6)
    assert tuple(out.shape) == (3, 1)
