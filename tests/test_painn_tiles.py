"""Host logic of the LDS-tile message kernel (gcnn_keras_amd/fused_painn.py::message_tile_table): every receiver in exactly
one tile, a tile inside one graph, edge ranges taken from the CSR, LDS limit respected.  CPU only (the LDS size comes from
the library's host-side entry point, no device call)."""
import numpy as np

from gcnn_keras_amd import synth
from gcnn_keras_amd.fused_painn import message_tile_table


def _csr(batch):
    ns, es = batch["node_splits"], batch["edge_splits"]
    n = int(ns[-1])
    recv = np.concatenate([batch["edge_indices"][es[g]:es[g + 1], 0] + ns[g] for g in range(len(ns) - 1)])
    assert np.all(np.diff(recv) >= 0)
    return np.searchsorted(recv, np.arange(n + 1), side="left").astype(np.int32)


def test_tile_table_partitions_receivers_inside_graphs():
    for batch, per in ((synth.md17_like_batch(num_graphs=7, seed=3), None), (synth.qm9_like_batch(num_graphs=40, seed=5), 6),
                       (synth.qm9_like_batch(num_graphs=9, seed=6), 1), (synth.qm9_like_batch(num_graphs=9, seed=6), 62)):
        ns, ptr = batch["node_splits"], _csr(batch)
        tl = message_tile_table(ns, ptr, 20, per=per)
        t = tl["table"]
        assert t.shape == (tl["count"], 8) and t.dtype == np.int32
        covered = np.concatenate([np.arange(r[0], r[1]) for r in t])
        assert np.array_equal(covered, np.arange(int(ns[-1])))               # each receiver once, in order
        for r_lo, r_hi, s_lo, s_hi, e_lo, e_hi, _, _ in t:
            g = int(np.searchsorted(ns, r_lo, side="right") - 1)
            assert ns[g] == s_lo and ns[g + 1] == s_hi and s_lo <= r_lo < r_hi <= s_hi
            assert e_lo == ptr[r_lo] and e_hi == ptr[r_hi] and r_hi - r_lo <= 62
        assert tl["max_rows"] == int(np.max(np.diff(ns))) and tl["max_edges"] == int(np.max(t[:, 5] - t[:, 4]))


def test_tile_table_refuses_what_does_not_fit_lds_and_handles_empty_graphs():
    b = synth.qm9_like_batch(num_graphs=5, seed=8)
    ns = np.concatenate([b["node_splits"][:3], [b["node_splits"][2]], b["node_splits"][3:]])   # an empty graph inside
    tl = message_tile_table(ns, _csr(b), 20)
    assert np.array_equal(np.concatenate([np.arange(r[0], r[1]) for r in tl["table"]]), np.arange(int(ns[-1])))
    big = np.array([0, 80], dtype=np.int64)                                    # 80 nodes x 3 KB of rows: beyond 160 KB
    assert message_tile_table(big, np.zeros(81, np.int32), 20) is None
    assert message_tile_table(np.array([0, 0]), np.zeros(1, np.int32), 20) is None
    assert message_tile_table(b["node_splits"], _csr(b), 32) is None          # no free bias slot
