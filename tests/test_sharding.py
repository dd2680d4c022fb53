"""Graph sharding + prediction all-gather with world_size 2 on the CPU (gloo).  The per-shard forward is played by the
oracle here (tests may use it as the checker / stand-in; the product forward needs an MI355X)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gcnn_keras_amd import sharding, synth
from oracle import kgcnn_oracle as ko


def test_shard_bounds_balance_and_cover():
    b = synth.qm9_like_batch(num_graphs=40, seed=3)
    for world in (1, 2, 3, 8):
        bounds = sharding.shard_bounds_by_edges(b["edge_splits"], world)
        assert bounds[0][0] == 0 and bounds[-1][1] == 40
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
        edges = [int(b["edge_splits"][hi] - b["edge_splits"][lo]) for lo, hi in bounds]
        assert sum(edges) == int(b["edge_splits"][-1])
        if world > 1:
            assert max(edges) - min(edges) <= 2 * int(np.diff(b["edge_splits"]).max())


def test_take_shard_rebases_partitions():
    b = synth.qm9_like_batch(num_graphs=10, seed=4)
    s = sharding.take_shard(b, 3, 7)
    assert s["node_splits"][0] == 0 and s["edge_splits"][0] == 0
    assert len(s["node_number"]) == s["node_splits"][-1] and len(s["edge_indices"]) == s["edge_splits"][-1]
    n0 = b["node_splits"][3]
    assert np.array_equal(s["node_coordinates"], b["node_coordinates"][n0:n0 + s["node_splits"][-1]])
    # sample indices are unchanged and stay inside their graphs
    for g in range(4):
        blk = s["edge_indices"][s["edge_splits"][g]:s["edge_splits"][g + 1]]
        assert blk.size == 0 or blk.max() < s["node_splits"][g + 1] - s["node_splits"][g]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = synth.qm9_like_batch(num_graphs=9, seed=21)
    p = synth.schnet_params(seed=7, random_bias=True)
    shard, bounds = sharding.shard_batch(b, rank, world)
    pred = ko.schnet_forward(p, ko.R(shard["node_number"], shard["node_splits"]),
                             ko.R(shard["node_coordinates"], shard["node_splits"]),
                             ko.R(shard["edge_indices"], shard["edge_splits"]), depth=3)
    full = sharding.all_gather_predictions(torch.from_numpy(pred), bounds)
    if rank == 0:
        queue.put(full.numpy())
    dist.destroy_process_group()


def test_sharded_forward_matches_whole_batch_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = queue.get(timeout=180)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    b = synth.qm9_like_batch(num_graphs=9, seed=21)
    p = synth.schnet_params(seed=7, random_bias=True)
    ref = ko.schnet_forward(p, ko.R(b["node_number"], b["node_splits"]), ko.R(b["node_coordinates"], b["node_splits"]),
                            ko.R(b["edge_indices"], b["edge_splits"]), depth=3)
    assert got.shape == ref.shape == (9, 1)
    assert np.max(np.abs(got - ref)) <= 1e-6 * np.max(np.abs(ref))   # graphs are independent: same rows, same math
