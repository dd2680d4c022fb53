"""On-GPU SetRange (csrc/mp_radius.hip) vs the host restatement of define_adjacency_from_distance, which is itself
pinned to edge lists produced by the reference's function (tests/test_synth.py)."""
import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_distance,max_neighbours", [(4.0, 30), (5.0, 10000), (4.0, 5), (2.0, 3), (None, 4)])
def test_set_range_matches_reference_rule(max_distance, max_neighbours, golden_dir):
    from gcnn_keras_amd.graph.preprocessor import SetRange
    from gcnn_keras_amd.ragged import RaggedTensor
    b = synth.qm9_like_batch(num_graphs=40, seed=77)
    xyz = RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"])
    idx, dist = SetRange(max_distance=max_distance, max_neighbours=max_neighbours)(xyz)
    got_rows, got_d = idx.numpy_rows(), dist.numpy_rows()
    for g in range(40):
        pts = b["node_coordinates"][b["node_splits"][g]:b["node_splits"][g + 1]]
        ref = synth.radius_graph(pts, max_distance=max_distance, max_neighbours=max_neighbours)
        assert np.array_equal(got_rows[g], ref), g
        if len(ref):
            d_ref = synth.distance_matrix(pts)[ref[:, 0], ref[:, 1]]
            assert np.array_equal(got_d[g][:, 0], d_ref.astype(np.float32))
    assert idx.values.dtype == torch.int64


def test_set_range_golden_cases_and_pipeline(golden_dir):
    """The edge lists made by the reference's own function (tests/golden/radius_graph_cases.npz), then straight into
    GatherNodes / PoolingLocalEdges."""
    import os
    from gcnn_keras_amd.graph.preprocessor import SetRange
    from gcnn_keras_amd.layers.pooling import PoolingLocalEdges
    from gcnn_keras_amd.ragged import RaggedTensor
    d = np.load(os.path.join(golden_dir, "radius_graph_cases.npz"))
    for c in range(6):
        md, mn = d["args_%d" % c]
        pts = d["xyz_%d" % c]
        xyz = RaggedTensor.from_numpy(pts, np.array([0, len(pts)], np.int64))
        idx, dist = SetRange(max_distance=float(md), max_neighbours=int(mn))(xyz)
        assert np.array_equal(idx.values.cpu().numpy(), d["edges_%d" % c]), c
    ones = RaggedTensor(torch.ones((idx.values.shape[0], 4), device="cuda"), idx.row_splits)
    nodes = RaggedTensor(torch.zeros((len(pts), 4), device="cuda"), xyz.row_splits)
    deg = PoolingLocalEdges("sum")([nodes, ones, idx]).values[:, 0].cpu().numpy()
    assert np.array_equal(deg, np.bincount(d["edges_5"][:, 0], minlength=len(pts)).astype(np.float32))
