"""Pins the CPU oracle to the reference's own known answers."""
import os

import numpy as np

from oracle import kgcnn_oracle as ko


def test_partition_row_indexing_docstring_example():
    # kgcnn/ops/partition.py:112-120 (index lengths [3, 1]; the docstring's [3, 2] is a typo)
    out = ko.partition_row_indexing(np.array([0, 0, 1, 1]), np.array([2, 2]), np.array([3, 1]),
                                    "row_lengths", "row_lengths")
    assert out.tolist() == [0, 0, 1, 3]
    assert np.array([10, 20, 30, 40])[out].tolist() == [10, 10, 20, 40]
    # inverse direction and identity (partition.py:135-137, :156-157)
    back = ko.partition_row_indexing(out, np.array([2, 2]), np.array([3, 1]), "row_lengths", "row_lengths",
                                     from_indexing="batch", to_indexing="sample")
    assert back.tolist() == [0, 0, 1, 1]
    same = ko.partition_row_indexing(out, np.array([2, 2]), np.array([3, 1]), "row_lengths", "row_lengths",
                                     from_indexing="batch", to_indexing="batch")
    assert same.tolist() == out.tolist()


def test_gather_nodes_reference_case(golden_dir):
    # test/test_gather.py:28-44
    d = np.load(os.path.join(golden_dir, "gather_case.npz"))
    node = ko.ragged_from_row_lengths(np.concatenate([d["n0"], d["n1"]]), [len(d["n0"]), len(d["n1"])])
    idx = ko.ragged_from_row_lengths(np.concatenate([d["ei0"], d["ei1"]]), [len(d["ei0"]), len(d["ei1"])])
    g = ko.gather_nodes(node, idx)
    np_gather = np.reshape(d["n1"][d["ei1"]], (28, 2 * 1))
    assert np.sum(np.abs(ko.ragged_rows(g)[1] - np_gather)) < 1e-6
    g2 = ko.gather_nodes(node, idx, concat_axis=None)
    assert ko.ragged_rows(g2)[1].shape == (28, 2, 1)
    assert np.sum(np.abs(ko.ragged_rows(g2)[1] - d["n1"][d["ei1"]])) < 1e-6


def test_attention_pooling_known_answer():
    # test/test_conv_attention.py:34-43
    nodes = ko.ragged_from_row_lengths(np.array([[1.0], [1.0]]), [1, 1])
    edges = ko.ragged_from_row_lengths(np.array([[100.0], [0.0], [100.0], [0.0]]), [2, 2])
    att = ko.ragged_from_row_lengths(np.array([[0.0], [1.0], [0.0], [1.0]]), [2, 2])
    idx = ko.ragged_from_row_lengths(np.zeros((4, 2), dtype=np.int64), [2, 2])
    res = ko.pooling_local_edges_attention(nodes, edges, att, idx)
    assert abs(ko.ragged_rows(res)[0][0, 0] - 100.0 / (np.exp(1) + 1)) < 1e-4


def test_bessel_basis_reference_asset(golden_dir):
    # test/test_geom.py:79-128 + test/assets/bessel_basis_reference.npz (data copied into tests/golden)
    d = np.load(os.path.join(golden_dir, "bessel_basis_reference.npz"))
    x = ko.ragged_from_row_lengths(np.concatenate([d["x0"], d["x1"]]).astype(np.float32), [5, 11])
    ei = ko.ragged_from_row_lengths(np.concatenate([d["ei0"], d["ei1"]]), [20, 108])
    a, b = ko.node_position(x, ei)
    dist = ko.node_distance_euclidean(a, b)
    bes = ko.ragged_rows(ko.bessel_basis(dist, 10, 5.0))
    assert bes[0].shape == (20, 10) and bes[1].shape == (108, 10)
    assert np.max(np.abs(d["bessel_basis_0"] - bes[0])) < 1e-5
    assert np.max(np.abs(d["bessel_basis_1"] - bes[1])) < 1e-5


def test_lstm_pool_shape_case_segment_rows():
    # test/test_pool_pooling.py:30-39 pins only the output row count (8 nodes in graph 0)
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "gather_case.npz"))
    node = ko.ragged_from_row_lengths(np.concatenate([d["n0"], d["n1"]]), [8, 15])
    idx = ko.ragged_from_row_lengths(np.concatenate([d["ei0"], d["ei1"]]), [14, 28])
    edges = ko.ragged_from_row_lengths(np.ones((42, 3), np.float32), [14, 28])
    out = ko.pooling_local_edges(node, edges, idx, "sum")
    assert ko.ragged_rows(out)[0].shape == (8, 3)


def test_dmpnn_gather_edges_pairs_reference_case():
    # reference test/test_conv_dmpnn.py:11-28: edges gathered at their reverse pairs; the reverse-pair indices of
    # ei1[0] = [[0,1],[1,0],[1,2],[2,1]] are [1, 0, 3, 2] (set_edge_indices_reverse)
    e1 = [np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0], [3.0, 3.0], [4.0, 4.0]]),
          np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 2.0], [3.0, 3.0]])]
    pairs = [np.array([[1], [0], [3], [2], [-1]]), np.array([[-1], [2], [1], [-1]])]
    edges = ko.ragged_from_list(e1, np.float32, (2,))
    pair_index = ko.ragged_from_list(pairs, np.int64, (1,))
    out = ko.ragged_rows(ko.dmpnn_gather_edges_pairs(edges, pair_index))
    assert np.max(np.abs(out[0][:4] - np.array([[1.0, 1.0], [0.0, 0.0], [3.0, 3.0], [2.0, 2.0]]))) < 1e-4
    assert np.array_equal(out[0][4], [0.0, 0.0])                      # no reverse edge -> zeros (dmpnn_conv.py:44-46)
    assert np.array_equal(out[1], [[0.0, 0.0], [2.0, 2.0], [1.0, 1.0], [0.0, 0.0]])
