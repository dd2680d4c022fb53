"""BASELINE config 4 on one GPU: the bench's shard builder (node data from the seeded host generator, edge counts and
edge lists from the engine's on-GPU SetRange, shards balanced by edge count) feeds the fused HIP forward, and the
predictions go through ``sharding.all_gather_predictions`` under a world-size-1 RCCL ("nccl") process group - the same
code path every rank runs on an 8-GPU node, minus the peers."""
import socket

import numpy as np
import pytest
import torch

from gcnn_keras_amd import sharding, synth
from helpers import mol_inputs
from oracle import kgcnn_oracle as ko
from parity import assert_rows_close

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    from gcnn_keras_amd.literature import Schnet
    p = synth.schnet_params(seed=7, random_bias=True)
    model = Schnet.make_model(depth=3)
    model.set_weights(list(p.values()))
    return model, p


def test_config4_shards_cover_the_batch_and_match_the_whole_forward():
    import bench
    total, world = 300, 3
    model, p = _model()
    whole = synth.qm9_like_batch(num_graphs=total, seed=3456)      # same node stream; edges by the host rule
    want = model(mol_inputs(whole)).cpu().numpy()
    parts, edges = [], 0
    for rank in range(world):                                      # the ranks of a node, played one after the other
        inputs, bounds, total_edges = bench.build_config4_shard(total, rank, world)
        assert total_edges == int(whole["edge_splits"][-1])        # on-GPU SetRange counts == reference rule on the host
        lo, hi = bounds[rank]
        assert inputs[0].nrows() == hi - lo
        e0, e1 = int(whole["edge_splits"][lo]), int(whole["edge_splits"][hi])
        assert np.array_equal(inputs[2].values.cpu().numpy(), whole["edge_indices"][e0:e1])   # bit-identical edge lists
        edges += int(inputs[2].values.shape[0])
        assert model.fused.accepts(inputs)
        parts.append(model(inputs).cpu().numpy())
        assert model.fused.last == "direct"
    assert edges == total_edges and bounds[0][0] == 0 and bounds[-1][1] == total
    sizes = [int(whole["edge_splits"][hi] - whole["edge_splits"][lo]) for lo, hi in bounds]
    assert max(sizes) - min(sizes) <= 2 * int(np.diff(whole["edge_splits"]).max())            # balanced by edge count
    got = np.concatenate(parts, axis=0)
    assert got.shape == want.shape == (total, 1)
    assert np.max(np.abs(got - want)) <= 2e-6 * np.max(np.abs(want))   # graphs are independent (tile boundaries move)
    ref = ko.schnet_forward(p, ko.R(whole["node_number"], whole["node_splits"]),
                            ko.R(whole["node_coordinates"], whole["node_splits"]),
                            ko.R(whole["edge_indices"], whole["edge_splits"]), depth=3)
    assert_rows_close(got, ref, what="sharded forward vs oracle")


def test_hip_forward_all_gather_under_rccl_world_size_1():
    import torch.distributed as dist
    model, p = _model()
    b = synth.qm9_like_batch(num_graphs=40, seed=21)
    shard, bounds = sharding.shard_batch(b, 0, 1)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        before = model.get_weights()
        sent = sharding.broadcast_weights(model, src=0)            # one flat ncclBroadcast, written back in place
        assert sent == sum(w.size for w in before) and all(np.array_equal(a, c) for a, c in zip(before, model.get_weights()))
        pred = model(mol_inputs(shard))
        full = sharding.all_gather_predictions(pred, bounds)       # all_gather_into_tensor over RCCL
        torch.cuda.synchronize()
        assert full.is_cuda and tuple(full.shape) == (40, 1)
        assert torch.equal(full, pred)
    finally:
        dist.destroy_process_group()
    ref = ko.schnet_forward(p, ko.R(b["node_number"], b["node_splits"]), ko.R(b["node_coordinates"], b["node_splits"]),
                            ko.R(b["edge_indices"], b["edge_splits"]), depth=3)
    assert_rows_close(full.cpu().numpy(), ref, what="gathered predictions")
