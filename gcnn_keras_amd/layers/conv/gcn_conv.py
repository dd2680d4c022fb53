"""GCN layer (mirror of kgcnn/layers/conv/gcn_conv.py:10-103): Dense -> gather sender rows -> weighted segment
reduce at the receiver -> activation.  The sparse-adjacency-matmul variant is unused by kgcnn.literature.GCN and out
of scope."""
from ..base import GraphBaseLayer
from ..gather import GatherNodesOutgoing
from ..modules import Activation, Dense
from ..pooling import PoolingWeightedLocalEdges


class GCN(GraphBaseLayer):
    r""":math:`\sigma(A_s (XW + b))` with the pre-scaled adjacency given as edge weights ``(batch, [M], 1)``."""

    def __init__(self, units, pooling_method="sum", normalize_by_weights=False, activation="kgcnn>leaky_relu",
                 use_bias=True, kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None,
                 kernel_constraint=None, bias_constraint=None, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        self.normalize_by_weights = normalize_by_weights
        self.pooling_method = pooling_method
        self.units = units
        kernel_args = {"kernel_regularizer": kernel_regularizer, "activity_regularizer": activity_regularizer,
                       "bias_regularizer": bias_regularizer, "kernel_constraint": kernel_constraint,
                       "bias_constraint": bias_constraint, "kernel_initializer": kernel_initializer,
                       "bias_initializer": bias_initializer, "use_bias": use_bias}
        pool_args = {"pooling_method": pooling_method, "normalize_by_weights": normalize_by_weights}
        # NB: like the reference (kgcnn/layers/base.py:9-11) is_sorted / has_unconnected are not handed to sub-layers
        self.lay_gather = GatherNodesOutgoing()
        self.lay_dense = Dense(units=self.units, activation="linear", **kernel_args)
        self.lay_pool = PoolingWeightedLocalEdges(**pool_args)
        self.lay_act = Activation(activation)

    def build(self, input_shape):
        super().build(input_shape)
        self.lay_dense.ensure_built(tuple(input_shape[0]))

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes (batch,[N],F), edge weights (batch,[M],1), edge_index (batch,[M],2)]``."""
        node, edges, edge_index = inputs
        no = self.lay_dense(node, **kwargs)
        fused = self._fused_aggregate(node, no, edges, edge_index)
        if fused is not None:
            return fused
        no = self.lay_gather([no, edge_index], **kwargs)
        nu = self.lay_pool([node, no, edge_index, edges], **kwargs)
        out = self.lay_act(nu, **kwargs)
        return out

    def _fused_aggregate(self, node, no, edges, edge_index):
        """gather -> weighted pool -> activation in one kernel (``mp_gather_segment_reduce_csr_f32``) when the
        configuration allows: scalar edge weights, an engine activation, no gradient requested."""
        import torch
        from ... import _ffi
        from ...autograd import needs_grad
        from ...ops.segment import reduce_op_code
        act = self.lay_act.activation
        if act == "softmax" or act not in _ffi.ACTIVATION_CODES or needs_grad(no.values, edges.values):
            return None
        w = edges.values
        if w.dim() != 2 or int(w.shape[1]) != 1 or no.values.dim() != 2:
            return None
        plan = edge_index.index_plan(node)
        pool = self.lay_pool
        ptr, perm, seg = plan.csr(pool.pooling_index, assume_sorted=pool.is_sorted)
        n_out = plan.N
        if not pool.has_unconnected:
            n_out = int(seg[-1].item()) + 1 if plan.M > 0 else 0
        x = no.values.contiguous()
        out = torch.empty((n_out, int(x.shape[1])), dtype=torch.float32, device=x.device)
        _ffi.call("mp_gather_segment_reduce_csr_f32", reduce_op_code(pool.pooling_method), _ffi.ptr(x), plan.N,
                  int(x.shape[1]), _ffi.ptr(plan.col(1).contiguous()), plan.M, _ffi.ptr(ptr), _ffi.ptr(perm), n_out,
                  _ffi.ptr(w.contiguous().view(-1)), 1 if pool.normalize_by_weights else 0,
                  _ffi.activation_code(act), 0.05, _ffi.ptr(out), _ffi.stream())
        return node.with_values(out)

    def get_config(self):
        config = super().get_config()
        config.update({"normalize_by_weights": self.normalize_by_weights, "pooling_method": self.pooling_method,
                       "units": self.units})
        conf_dense = self.lay_dense.get_config()
        for x in ["kernel_regularizer", "activity_regularizer", "bias_regularizer", "kernel_constraint",
                  "bias_constraint", "kernel_initializer", "bias_initializer", "use_bias"]:
            config.update({x: conf_dense[x]})
        conf_act = self.lay_act.get_config()
        config.update({"activation": conf_act["activation"]})
        return config
