"""Pooling / aggregation layers (mirror of kgcnn/layers/pooling.py) on the HIP engine.

The reference's ``argsort -> gather -> segment_op -> scatter_nd`` chain (4 passes over the (M, F) messages) is one
receiver-parallel kernel over a CSR that is built once per batch (see csrc/mp_segment.hip).
"""
import torch

from .. import _ffi
from ..ops.segment import reduce_op_code, segment_reduce_csr, segment_softmax_csr
from ..ops.scatter import scatter_op_code
from .base import GraphBaseLayer


def _pool_local(layer, nodes, edges, idx, op, weights=None, normalize=False):
    plan = idx.index_plan(nodes)
    if layer.ragged_validate:
        plan.validate()
    ptr, perm, seg = plan.csr(layer.pooling_index, assume_sorted=layer.is_sorted)
    n_out = plan.N
    if not layer.has_unconnected:
        # reference: rows = max(receiver) + 1 (kgcnn/layers/pooling.py:71-76 without the scatter_nd pad)
        n_out = int(seg[-1].item()) + 1 if plan.M > 0 else 0
    w = None if weights is None else weights.values
    out = segment_reduce_csr(op, edges.values, ptr, perm, n_out, weight=w, normalize_by_weight=normalize,
                             seg_ids=plan.col(layer.pooling_index))
    return nodes.with_values(out)


class PoolingLocalEdges(GraphBaseLayer):
    r"""Aggregate edge embeddings at the receiving node ``i = idx[:, pooling_index]``
    (kgcnn/layers/pooling.py:11-88).  **Default ``pooling_method`` is "mean"** like the reference."""

    def __init__(self, pooling_method="mean", pooling_index=0, **kwargs):
        super().__init__(**kwargs)
        self.pooling_method = pooling_method
        self.pooling_index = pooling_index

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes (batch,[N],F), edges (batch,[M],F), tensor_index (batch,[M],2)]``."""
        self.assert_ragged_input_rank(list(inputs))
        nodes, edges, idx = inputs
        return _pool_local(self, nodes, edges, idx, reduce_op_code(self.pooling_method))

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method, "pooling_index": self.pooling_index})
        return config


PoolingLocalMessages = PoolingLocalEdges


class PoolingWeightedLocalEdges(GraphBaseLayer):
    r"""Weighted aggregation (kgcnn/layers/pooling.py:92-182): ``edges * weights`` is formed BEFORE the reduce for
    every method; ``normalize_by_weights`` divides by the segment sum of weights with ``divide_no_nan``."""

    def __init__(self, pooling_method="mean", normalize_by_weights=False, pooling_index=0, **kwargs):
        super().__init__(**kwargs)
        self.pooling_method = pooling_method
        self.normalize_by_weights = normalize_by_weights
        self.pooling_index = pooling_index

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes, edges, tensor_index, weights (batch,[M],1)]``."""
        self.assert_ragged_input_rank(list(inputs))
        nodes, edges, idx, weights = inputs
        return _pool_local(self, nodes, edges, idx, reduce_op_code(self.pooling_method), weights=weights,
                           normalize=self.normalize_by_weights)

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method, "normalize_by_weights": self.normalize_by_weights,
                       "pooling_index": self.pooling_index})
        return config


PoolingWeightedLocalMessages = PoolingWeightedLocalEdges


def _pool_graph(x, op, weights=None):
    """Per-graph reduce of ragged ``x``; returns a dense tensor with ``max(rowid) + 1`` rows like
    ``tf.math.segment_*`` on ``value_rowids`` (kgcnn/layers/pooling.py:215-219): trailing empty graphs are dropped,
    interior empty graphs give 0."""
    _ffi.require_device(x.values, x.row_splits)
    vals = x.values.contiguous()
    g = x.nrows()
    from ..autograd import PoolGraph, needs_grad
    if needs_grad(vals):
        if weights is not None:
            raise NotImplementedError("gradient of weighted graph pooling is outside the force path")
        out = PoolGraph.apply(vals, op, x.row_splits, g)
        splits = x.row_splits_host()
        rows = g
        while rows > 0 and splits[rows] == splits[rows - 1]:
            rows -= 1
        return out[:rows] if rows != g else out
    elems = 1
    for d in vals.shape[1:]:
        elems *= int(d)
    out = torch.empty((g,) + tuple(vals.shape[1:]), dtype=torch.float32, device=vals.device)
    w = None if weights is None else weights.values.contiguous().view(-1)
    _ffi.call("mp_pool_graph_f32", op, _ffi.ptr(vals), _ffi.ptr(x.row_splits), g, max(elems, 1), _ffi.ptr(w),
              _ffi.ptr(out), _ffi.stream())
    splits = x.row_splits_host()
    rows = g
    while rows > 0 and splits[rows] == splits[rows - 1]:
        rows -= 1
    return out[:rows] if rows != g else out


class PoolingEmbedding(GraphBaseLayer):
    """Pool all nodes (or edges) of each graph to a graph embedding tensor (kgcnn/layers/pooling.py:186-229)."""

    def __init__(self, pooling_method="mean", **kwargs):
        super().__init__(**kwargs)
        self.pooling_method = pooling_method

    def call(self, inputs, **kwargs):
        inputs = self.assert_ragged_input_rank(inputs)
        return _pool_graph(inputs, reduce_op_code(self.pooling_method))

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method})
        return config


PoolingNodes = PoolingEmbedding
PoolingGlobalEdges = PoolingEmbedding


class PoolingWeightedEmbedding(GraphBaseLayer):
    """Weighted per-graph pooling (kgcnn/layers/pooling.py:233-284)."""

    def __init__(self, pooling_method="mean", **kwargs):
        super().__init__(**kwargs)
        self.pooling_method = pooling_method

    def call(self, inputs, **kwargs):
        nodes, weights = self.assert_ragged_input_rank(list(inputs))
        return _pool_graph(nodes, reduce_op_code(self.pooling_method), weights=weights)

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method})
        return config


PoolingWeightedNodes = PoolingWeightedEmbedding
PoolingWeightedGlobalEdges = PoolingWeightedEmbedding


class PoolingLocalEdgesAttention(GraphBaseLayer):
    r"""Attention pooling ``n_i = sum_j softmax_j(a_ij) e_ij`` (kgcnn/layers/pooling.py:464-546)."""

    def __init__(self, pooling_index=0, **kwargs):
        super().__init__(**kwargs)
        self.pooling_index = pooling_index

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes, edges, attention (batch,[M],1), edge_indices]``."""
        self.assert_ragged_input_rank(list(inputs))
        nodes, edges, attention, idx = inputs
        plan = idx.index_plan(nodes)
        ptr, perm, seg = plan.csr(self.pooling_index, assume_sorted=self.is_sorted)
        ats = segment_softmax_csr(attention.values, ptr, perm, plan.N)
        n_out = plan.N
        if not self.has_unconnected:
            n_out = int(seg[-1].item()) + 1 if plan.M > 0 else 0
        out = segment_reduce_csr(_ffi.MP_SUM, edges.values, ptr, perm, n_out, weight=ats)
        return nodes.with_values(out)

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_index": self.pooling_index})
        return config


class PoolingEmbeddingAttention(GraphBaseLayer):
    r"""Per-graph attention pooling ``s = sum_i softmax_i(a_i) n_i`` (kgcnn/layers/pooling.py:550-599)."""

    def call(self, inputs, **kwargs):
        nodes, attention = self.assert_ragged_input_rank(list(inputs))
        g = nodes.nrows()
        ptr = nodes.row_splits.to(torch.int32).contiguous()
        ats = segment_softmax_csr(attention.values, ptr, None, g)
        return _pool_graph(nodes, _ffi.MP_SUM, weights=nodes.with_values(ats))


PoolingNodesAttention = PoolingEmbeddingAttention


class RelationalPoolingLocalEdges(GraphBaseLayer):
    r"""Aggregate edges per (receiving node, relation) into ``(batch, [N], R, F)``
    (kgcnn/layers/pooling.py:603-675; unsorted scatter, ``tf.tensor_scatter_nd_*`` semantics on a zero tensor)."""

    def __init__(self, num_relations, pooling_method="sum", pooling_index=0, **kwargs):
        super().__init__(**kwargs)
        self.num_relations = num_relations
        self.pooling_method = pooling_method
        self.pooling_index = pooling_index

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes, edges, tensor_index, edge_relation (batch,[M])]``."""
        nodes, edges, idx, edge_rel = self.assert_ragged_input_rank(list(inputs))
        op = scatter_op_code(self.pooling_method)
        plan = idx.index_plan(nodes)
        recv = plan.col(self.pooling_index).contiguous()
        rel = edge_rel.values.to(torch.int32).contiguous()
        vals = edges.values.contiguous()
        elems = 1
        for d in vals.shape[1:]:
            elems *= int(d)
        out = torch.zeros((plan.N, self.num_relations) + tuple(vals.shape[1:]), dtype=torch.float32, device=vals.device)
        _ffi.call("mp_scatter_relational_f32", op, _ffi.ptr(vals), plan.M, max(elems, 1), _ffi.ptr(recv), _ffi.ptr(rel),
                  plan.N, self.num_relations, _ffi.ptr(out), _ffi.stream())
        return nodes.with_values(out)

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method, "pooling_index": self.pooling_index,
                       "num_relations": self.num_relations})
        return config
