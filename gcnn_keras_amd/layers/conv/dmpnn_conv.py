"""Directed message passing helpers of DMPNN (mirror of kgcnn/layers/conv/dmpnn_conv.py:9-89).

``DMPNNGatherEdgesPairs``: ``out[e] = edges[pair[e]]`` where ``pair[e]`` is the (per-graph) position of the reverse edge,
zeros where there is none (``pair[e] < 0``, dmpnn_conv.py:39-46).  ``DMPNNPPoolingEdgesDirected`` is the reference's
sequence (dmpnn_conv.py:83-87): sum the edge states at their receiver, gather that sum at each edge's sender, subtract the
state of the reverse edge.
"""
import torch

from ... import _ffi
from ..base import GraphBaseLayer
from ..gather import GatherNodesIngoing, GatherNodesOutgoing
from ..modules import LazySubtract, binary_values
from ..pooling import PoolingLocalEdges


class DMPNNGatherEdgesPairs(GraphBaseLayer):

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.gather_layer = GatherNodesIngoing()

    def call(self, inputs, **kwargs):
        """inputs: ``[edges (batch,[M],F), pair_index (batch,[M],1) int64]`` -> ``(batch,[M],F)``."""
        edges, pair_index = inputs
        pairs = pair_index.values
        has_pair = pairs >= 0
        safe = pair_index.with_values(torch.where(has_pair, pairs, torch.zeros_like(pairs)))
        gathered = self.gather_layer([edges, safe], **kwargs)
        mask = has_pair.to(gathered.values.dtype).reshape((-1,) + (1,) * (gathered.values.dim() - 1))
        return gathered.with_values(binary_values(_ffi.MP_MUL, gathered.values, mask))


class DMPNNPPoolingEdgesDirected(GraphBaseLayer):

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.pool_edge_1 = PoolingLocalEdges(pooling_method="sum")
        self.gather_edges = GatherNodesOutgoing()
        self.gather_pairs = DMPNNGatherEdgesPairs()
        self.subtract_layer = LazySubtract()

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes, edges (batch,[M],F), edge_index (batch,[M],2), edge_reverse_pair (batch,[M],1)]``."""
        nodes, edges, edge_index, reverse_pair = inputs
        received = self.pool_edge_1([nodes, edges, edge_index], **kwargs)
        at_sender = self.gather_edges([received, edge_index], **kwargs)
        return self.subtract_layer([at_sender, self.gather_pairs([edges, reverse_pair], **kwargs)], **kwargs)
