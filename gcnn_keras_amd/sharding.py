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


def all_gather_predictions(local_pred, bounds, group=None, empty_row=None):
    """Gather per-graph predictions of every rank into the original graph order.

    ``local_pred``: torch tensor ``(rows, L)`` on this rank's device.  The row count of rank r's part is taken from
    ``bounds`` (``hi - lo`` graphs), not from the tensor: shards differ in size, so every rank pads to the largest shard
    (equal chunks = ONE ``all_gather_into_tensor``) and the padding is dropped afterwards.

    A shard that ends in graphs without nodes returns fewer rows than it has graphs (``PoolingNodes`` follows
    ``tf.math.segment_*``, which sizes its output by the last non-empty segment, kgcnn/layers/pooling.py:215-218).  In the
    unsharded forward such graphs are gaps inside the batch - the segment op fills them with zeros and the output MLP maps
    that to a constant row - unless they end the whole batch, where they are dropped.  The gather reproduces both: rows a
    shard did not return are ``empty_row`` (the model's prediction for a graph without nodes, ``(L,)``; NaN when not given,
    so a missing value is never silently a number), and rows missing at the end of the LAST shard are dropped."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if len(bounds) != world:
        raise ValueError("all_gather_predictions: %d shard bounds for %d ranks" % (len(bounds), world))
    sizes = [hi - lo for lo, hi in bounds]
    if local_pred.dim() == 1:
        local_pred = local_pred.unsqueeze(-1)
    rows, width = int(local_pred.shape[0]), int(local_pred.shape[1])
    if rows > sizes[rank]:
        raise ValueError("all_gather_predictions: rank %d returned %d rows for %d graphs" % (rank, rows, sizes[rank]))
    chunk = max(sizes)
    # one extra column carries the number of rows the rank really returned (rides in the same collective)
    send = torch.zeros((chunk + 1, width), dtype=local_pred.dtype, device=local_pred.device)
    send[:rows] = local_pred
    send[chunk, 0] = rows
    recv = torch.empty((world * (chunk + 1), width), dtype=local_pred.dtype, device=local_pred.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.reshape(world, chunk + 1, width)
    returned = [int(v) for v in recv[:, chunk, 0].round().to(torch.int64).cpu().tolist()]
    parts = []
    for r in range(world):
        part = recv[r, :sizes[r]]
        if returned[r] < sizes[r]:
            if r == world - 1:
                part = part[:returned[r]]                      # trailing empty graphs of the batch: dropped, as unsharded
            else:
                part = part.clone()
                part[returned[r]:] = float("nan") if empty_row is None else torch.as_tensor(
                    empty_row, dtype=part.dtype, device=part.device).reshape(1, width)
        parts.append(part)
    return torch.cat(parts, dim=0)


def broadcast_weights(model, src=0, group=None):
    """Replicate the model's weights from rank ``src`` (one broadcast at init; ~1 MB for SchNet depth 3):
    all weight tensors travel as ONE flat buffer - one collective, latency-bound - and are written back in place, so the
    fused routes see moved version counters and refresh their packed images."""
    import torch
    import torch.distributed as dist
    tensors = [t for _, t in model.weights]
    if not tensors:
        return 0
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    if flat.is_cuda and dist.get_backend(group) == "gloo":     # CPU rehearsal of a GPU run: stage through the host
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src=src, group=group)
    at = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[at:at + n].reshape(t.shape))
            at += n
    return int(flat.numel())
