"""Parity at BASELINE.json's full sizes through size-independent properties (and the C oracle where it is fast enough).

* config 2 (128 graphs) and a config-4 shard (12 500 graphs, 2.5 M edges): graphs are independent, so the fused forward
  of the whole batch must equal, row for row, the forward of any sub-batch; permuting the graphs permutes the rows.
* segment-sum linearity and gather/scatter duality on the 2.5 M-edge index list.
* the C/OpenMP oracle port checks a 2 000-graph slice of the shard directly.
"""
import numpy as np
import pytest
import torch

from gcnn_keras_amd import sharding, synth
from oracle import kgcnn_oracle as ko
from parity import assert_rows_close, rowwise_rel

pytestmark = pytest.mark.gpu


def _fused(params, batch):
    from gcnn_keras_amd.engine import SchnetForward
    f = SchnetForward(params, depth=3, mode="fused")
    f.load_batch(batch)
    out = f.forward().cpu().numpy().copy()
    f.check_flags()
    return out


@pytest.fixture(scope="module")
def shard_batch():
    return synth.qm9_like_batch(num_graphs=12500, seed=3456)


def test_config4_shard_graph_independence_and_c_oracle(shard_batch):
    from oracle import c_oracle
    p = synth.schnet_params(seed=7, random_bias=True)
    b = shard_batch
    assert b["edge_splits"][-1] > 2_400_000
    whole = _fused(p, b)
    assert whole.shape == (12500, 1) and np.all(np.isfinite(whole))
    # sub-batches: same rows (bit-identical is not required: tile boundaries move, so atomics pair differently)
    for lo, hi in [(0, 128), (6000, 6500), (12400, 12500)]:
        part = _fused(p, sharding.take_shard(b, lo, hi))
        assert rowwise_rel(part, whole[lo:hi]) <= 2e-6
    if c_oracle.available():
        sub = sharding.take_shard(b, 3000, 5000)
        ref = c_oracle.schnet_forward(p, sub["node_number"], sub["node_coordinates"], sub["edge_indices"],
                                      sub["node_splits"], sub["edge_splits"], depth=3)
        assert_rows_close(whole[3000:5000], ref, what="config-4 shard rows 3000-5000 vs the C oracle")


def test_default_bench_launch_group_rows_at_full_size():
    """What ``bench.py`` times by default: a launch group of FIVE independent 128-graph batches (config 2 size, different
    molecules each).  The union has ~4.1 k edge tiles - the size at which ``cfconv_dispatch`` takes the 8-wave build by its
    rounds x cost rule - and ~720 node tiles (the eight-wave node chains persistent over three tiles per workgroup).  Every
    member's rows must equal a forward of its own (2e-6: boundary sums pair differently in the union) and the C oracle's."""
    from gcnn_keras_amd.literature import Schnet
    from gcnn_keras_amd.ragged import RaggedTensor
    from oracle import c_oracle
    p = synth.schnet_params(seed=7, random_bias=True)
    model = Schnet.make_model(depth=3)
    model.set_weights(list(p.values()))
    batches = [synth.qm9_like_batch(num_graphs=128, seed=1234 + k) for k in range(5)]
    tiles = sum((int(b["edge_splits"][-1]) + 31) // 32 for b in batches)
    assert 3072 <= tiles and 37 * ((tiles + 2047) // 2048) < 20 * ((tiles + 1023) // 1024)   # the 8-wave build's range
    ins = [[RaggedTensor.from_numpy(b["node_number"], b["node_splits"]),
            RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
            RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])] for b in batches]
    alone = [model(x).cpu().numpy() for x in ins]
    first = [t.cpu().numpy() for t in model.fused.call_group(ins)]            # direct launch
    assert model.fused.last == "direct"
    again = [t.cpu().numpy() for t in model.fused.call_group(ins)]            # the group's captured graph
    assert model.fused.last == "graph"
    for k, b in enumerate(batches):
        assert first[k].shape == (128, 1) and np.array_equal(first[k], again[k])
        assert rowwise_rel(first[k], alone[k]) <= 2e-6
        if c_oracle.available():
            ref = c_oracle.schnet_forward(p, b["node_number"], b["node_coordinates"], b["edge_indices"], b["node_splits"],
                                          b["edge_splits"], depth=3)
            assert_rows_close(first[k], ref, what="default bench group, member %d vs the C oracle" % k)
    model.fused.check_flags()
    # "several launch sequences in flight" (flag bit 9, what engine.SchnetForward sets for in_flight > 1): the union's node
    # chains run on 128 persistent workgroups instead of 256 - other workgroups take the tiles, the arithmetic per tile is
    # the same, so the rows are the same bits
    half = Schnet.make_model(depth=3)
    half.set_weights(list(p.values()))
    half.fused.cfconv_flags |= 512
    got = [t.cpu().numpy() for t in half.fused.call_group(ins)]
    for k in range(5):
        assert np.array_equal(got[k], first[k])
    half.fused.check_flags()


def test_config2_graph_permutation_equivariance():
    p = synth.schnet_params(seed=7, random_bias=True)
    b = synth.qm9_like_batch(num_graphs=128, seed=1234)
    base = _fused(p, b)
    order = np.random.default_rng(0).permutation(128)
    parts = [sharding.take_shard(b, int(g), int(g) + 1) for g in order]
    pb = {"node_number": np.concatenate([q["node_number"] for q in parts]),
          "node_coordinates": np.concatenate([q["node_coordinates"] for q in parts]),
          "edge_indices": np.concatenate([q["edge_indices"] for q in parts]),
          "node_splits": np.concatenate([[0], np.cumsum([q["node_splits"][-1] for q in parts])]).astype(np.int64),
          "edge_splits": np.concatenate([[0], np.cumsum([q["edge_splits"][-1] for q in parts])]).astype(np.int64)}
    perm_out = _fused(p, pb)
    assert np.max(np.abs(perm_out - base[order])) <= 2e-6 * np.max(np.abs(base))


def test_segment_sum_linearity_and_duality_at_full_size(shard_batch):
    """sum-pooling is linear; <pool(e), n> == <e, gather_in(n)> (gather and segment-sum are adjoint) on 2.5 M edges."""
    from gcnn_keras_amd.layers.gather import GatherNodesIngoing
    from gcnn_keras_amd.layers.pooling import PoolingLocalEdges
    from gcnn_keras_amd.ragged import RaggedTensor
    b = shard_batch
    n, m, f = int(b["node_splits"][-1]), int(b["edge_splits"][-1]), 16
    g = torch.Generator(device="cuda").manual_seed(0)
    nodes = RaggedTensor(torch.randn(n, f, device="cuda", generator=g), torch.from_numpy(b["node_splits"]).cuda())
    idx = RaggedTensor(torch.from_numpy(b["edge_indices"]).cuda(), torch.from_numpy(b["edge_splits"]).cuda())
    e1 = RaggedTensor(torch.randn(m, f, device="cuda", generator=g), idx.row_splits)
    e2 = RaggedTensor(torch.randn(m, f, device="cuda", generator=g), idx.row_splits)
    pool = PoolingLocalEdges("sum")
    p1, p2 = pool([nodes, e1, idx]).values, pool([nodes, e2, idx]).values
    p12 = pool([nodes, e1.with_values(e1.values + e2.values), idx]).values
    assert float((p12 - (p1 + p2)).abs().max()) <= 1e-5 * float(p12.abs().max())
    gathered = GatherNodesIngoing()([nodes, idx]).values
    lhs = float((p1.double() * nodes.values.double()).sum())
    rhs = float((e1.values.double() * gathered.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0) + 1e-3
    # every edge lands on exactly one node: column sums agree
    assert float((p1.double().sum(0) - e1.values.double().sum(0)).abs().max()) <= 1e-3
