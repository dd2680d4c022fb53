"""Pins the synthetic-input generator to edge lists made by the reference's own NumPy functions."""
import os

import numpy as np

from gcnn_keras_amd import synth


def test_radius_graph_matches_reference_rule(golden_dir):
    d = np.load(os.path.join(golden_dir, "radius_graph_cases.npz"))
    n_cases = len([k for k in d.files if k.startswith("xyz_")])
    assert n_cases >= 6
    for c in range(n_cases):
        md, mn = d["args_%d" % c]
        ei = synth.radius_graph(d["xyz_%d" % c], max_distance=float(md), max_neighbours=int(mn))
        assert ei.dtype == np.int64
        assert np.array_equal(ei, d["edges_%d" % c]), c
        if len(ei) > 1:  # receiver-sorted, row-major
            assert np.all(np.diff(ei[:, 0]) >= 0)


def test_degree_sym_weights_match_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "gcn_weight_case.npz"))
    w = synth.rescale_edge_weights_degree_sym(d["edge_indices"], np.ones((len(d["edge_indices"]), 1), np.float32))
    assert np.array_equal(w, d["weights"])


def test_config_sizes():
    b = synth.qm9_like_batch()
    assert b["node_splits"][-1] == 2301 and b["edge_splits"][-1] == 26190  # BASELINE.md calibration
    assert len(b["node_splits"]) == 129
    b = synth.md17_like_batch()
    assert b["node_splits"][-1] == 1344 and b["edge_splits"][-1] == 20586
    t = synth.toy_batch()
    assert t["edge_indices"].tolist() == [[0, 1], [1, 0], [0, 1], [1, 2], [2, 0], [0, 0]]
