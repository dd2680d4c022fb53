"""Graph sharding across the GPUs of one node.

Batched molecular graphs are independent (the disjoint union has no edge between graphs), so the batch is cut into
contiguous graph ranges balanced by edge count, every rank runs the forward on its own shard with no exchange, and one
``all_gather`` (RCCL over xGMI on the GPUs, gloo in the CPU tests) returns the per-graph predictions in the original
order.  Per-graph "sample" indices need no rewriting (``node_indexing="sample"``, kgcnn/layers/base.py:27); only the
row_splits are rebased.  A single large graph (config 5) does not shard: replicas only.
"""
import numpy as np


def shard_bounds_by_edges(edge_splits, world_size):
    """Contiguous graph ranges ``[lo, hi)`` per rank with near-equal edge counts (prefix sum of edge row lengths)."""
    edge_splits = np.asarray(edge_splits, dtype=np.int64)
    g = len(edge_splits) - 1
    total = int(edge_splits[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        cut = int(np.searchsorted(edge_splits, target, side="left"))
        cut = min(max(cut, bounds[-1]), g)
        bounds.append(cut)
    bounds.append(g)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def take_shard(batch, lo, hi):
    """Sub-batch of graphs ``lo..hi-1`` with rebased row_splits (values are contiguous slices, indices unchanged)."""
    ns, es = np.asarray(batch["node_splits"]), np.asarray(batch["edge_splits"])
    n0, n1, e0, e1 = int(ns[lo]), int(ns[hi]), int(es[lo]), int(es[hi])
    out = {"node_splits": (ns[lo:hi + 1] - n0).astype(np.int64), "edge_splits": (es[lo:hi + 1] - e0).astype(np.int64)}
    for key, val in batch.items():
        if key in ("node_splits", "edge_splits"):
            continue
        val = np.asarray(val)
        if key.startswith("node"):
            out[key] = val[n0:n1]
        elif key.startswith("edge"):
            out[key] = val[e0:e1]
        else:
            out[key] = val
    return out


def shard_batch(batch, rank, world_size):
    bounds = shard_bounds_by_edges(batch["edge_splits"], world_size)
    lo, hi = bounds[rank]
    return take_shard(batch, lo, hi), bounds


def all_gather_predictions(local_pred, bounds, group=None):
    """Gather per-graph predictions of every rank into the original graph order.

    ``local_pred``: torch tensor ``(hi - lo, L)`` on this rank's device.  Shards differ in size, so each rank pads to
    the largest shard (equal chunks = one ``all_gather_into_tensor``), then the padding is dropped."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [hi - lo for lo, hi in bounds]
    width = int(local_pred.shape[1]) if local_pred.dim() > 1 else 1
    chunk = max(sizes)
    send = torch.zeros((chunk, width), dtype=local_pred.dtype, device=local_pred.device)
    send[:local_pred.shape[0]] = local_pred.reshape(-1, width)
    recv = torch.empty((world * chunk, width), dtype=local_pred.dtype, device=local_pred.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = [recv[r * chunk:r * chunk + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)
