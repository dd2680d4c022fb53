"""GIN / GINE aggregation (mirror of kgcnn/layers/conv/gin_conv.py:10-164) on the engine's gather / segment kernels.

``GIN``:  ``(1 + eps) h_i + pool_{j in N(i)} h_j``                       (gin_conv.py:65-69)
``GINE``: ``(1 + eps) h_i + pool_{j in N(i)} act(h_j + e_ij)``           (gin_conv.py:147-153)

The node MLP that follows is a separate layer in the reference and stays one here.  GIN's gather + pool runs as the
fused gather-reduce kernel (``mp_gather_segment_reduce_csr_f32``, the kernel behind the GCN layer) when no gradient is
requested: the ``(M, F)`` gathered rows are never materialised.
"""
import torch

from ... import _ffi
from ...autograd import needs_grad
from ...ops.segment import reduce_op_code
from ..base import GraphBaseLayer
from ..gather import GatherNodesOutgoing
from ..modules import Activation, LazyAdd, binary_values
from ..pooling import PoolingLocalEdges


class _GINBase(GraphBaseLayer):

    def __init__(self, pooling_method, epsilon_learnable, **kwargs):
        super().__init__(**kwargs)
        self.pooling_method = pooling_method
        self.epsilon_learnable = epsilon_learnable
        self.eps_k = self.add_weight("epsilon_k", (), initializer="zeros")   # scalar, zero-initialised (gin_conv.py:47-48)

    def _self_term(self, node):
        scale = (1.0 + self.eps_k).reshape((1,) * node.values.dim())
        return node.with_values(binary_values(_ffi.MP_MUL, node.values, scale))

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method, "epsilon_learnable": self.epsilon_learnable})
        return config


class GIN(_GINBase):

    def __init__(self, pooling_method="sum", epsilon_learnable=False, **kwargs):
        super().__init__(pooling_method, epsilon_learnable, **kwargs)
        self.lay_gather = GatherNodesOutgoing()
        self.lay_pool = PoolingLocalEdges(pooling_method=self.pooling_method)
        self.lay_add = LazyAdd()

    def _neighbour_sum_fused(self, node, edge_index):
        x = node.values
        if x.dim() != 2 or needs_grad(x, self.eps_k):
            return None
        plan = edge_index.index_plan(node)
        ptr, perm, _ = plan.csr(self.lay_pool.pooling_index, assume_sorted=self.lay_pool.is_sorted)
        x = x.contiguous()
        out = torch.empty((plan.N, int(x.shape[1])), dtype=torch.float32, device=x.device)
        _ffi.call("mp_gather_segment_reduce_csr_f32", reduce_op_code(self.pooling_method), _ffi.ptr(x), plan.N,
                  int(x.shape[1]), _ffi.ptr(plan.col(1).contiguous()), plan.M, _ffi.ptr(ptr), _ffi.ptr(perm), plan.N,
                  None, 0, _ffi.activation_code("linear"), 0.0, _ffi.ptr(out), _ffi.stream())
        return node.with_values(out)

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes (batch,[N],F), edge_index (batch,[M],2)]`` -> ``(batch,[N],F)``."""
        node, edge_index = inputs
        pooled = self._neighbour_sum_fused(node, edge_index)
        if pooled is None:
            pooled = self.lay_pool([node, self.lay_gather([node, edge_index], **kwargs), edge_index], **kwargs)
        return self.lay_add([self._self_term(node), pooled], **kwargs)


class GINE(_GINBase):

    def __init__(self, pooling_method="sum", epsilon_learnable=False, activation="relu", activity_regularizer=None,
                 **kwargs):
        super().__init__(pooling_method, epsilon_learnable, **kwargs)
        self.layer_gather = GatherNodesOutgoing()
        self.layer_pool = PoolingLocalEdges(pooling_method=self.pooling_method)
        self.layer_add = LazyAdd()
        self.layer_act = Activation(activation=activation, activity_regularizer=activity_regularizer)

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes (batch,[N],F), edge_index (batch,[M],2), edges (batch,[M],F)]`` -> ``(batch,[N],F)``."""
        node, edge_index, edges = inputs
        msg = self.layer_act(self.layer_add([self.layer_gather([node, edge_index], **kwargs), edges]))
        pooled = self.layer_pool([node, msg, edge_index], **kwargs)
        return self.layer_add([self._self_term(node), pooled], **kwargs)

    def get_config(self):
        config = super().get_config()
        act = self.layer_act.get_config()
        config.update({"activation": act["activation"], "activity_regularizer": act["activity_regularizer"]})
        return config
