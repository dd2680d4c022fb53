"""Attention heads of GAT / GATv2 (mirror of kgcnn/layers/conv/gat_conv.py:9-230) on the engine.

Both heads end in ``PoolingLocalEdgesAttention`` (segment softmax of the logits over each receiver's edges, then the
weighted segment sum - csrc/mp_segment.hip); they differ in how the logit is formed:

* GAT   (gat_conv.py:103-114): ``a_ij = act(a^T [W n_i || W n_j (|| e_ij)])``, messages ``W n_j``
* GATv2 (gat_conv.py:200-214): ``a_ij = a^T act(W_a [n_i || n_j (|| e_ij)])``, messages ``W n_j``
"""
from ..base import GraphBaseLayer
from ..gather import GatherNodesIngoing, GatherNodesOutgoing
from ..modules import Activation, Dense, LazyAverage, LazyConcatenate
from ..pooling import PoolingLocalEdgesAttention

_KERNEL_KEYS = ("kernel_regularizer", "activity_regularizer", "bias_regularizer", "kernel_constraint", "bias_constraint",
                "kernel_initializer", "bias_initializer")


class _AttentionHead(GraphBaseLayer):
    """Shared constructor surface and config of the two heads."""

    def __init__(self, units, use_edge_features, use_final_activation, has_self_loops, activation, use_bias,
                 kernel_args, **kwargs):
        super().__init__(**kwargs)
        self.units = int(units)
        self.use_edge_features = use_edge_features
        self.use_final_activation = use_final_activation
        self.has_self_loops = has_self_loops
        self.use_bias = use_bias
        self._kernel_args = kernel_args
        self.lay_linear_trafo = Dense(units, activation="linear", use_bias=use_bias, **kernel_args)
        self._make_logit_layers(activation)
        self.lay_gather_in = GatherNodesIngoing()
        self.lay_gather_out = GatherNodesOutgoing()
        self.lay_concat = LazyConcatenate(axis=-1)
        self.lay_pool_attention = PoolingLocalEdgesAttention()
        if use_final_activation:
            self.lay_final_activ = Activation(activation=activation)

    def _attend(self, node, messages, logits, edge_index, **kwargs):
        h = self.lay_pool_attention([node, messages, logits, edge_index], **kwargs)
        return self.lay_final_activ(h, **kwargs) if self.use_final_activation else h

    def _pair_features(self, left, right, edge, **kwargs):
        parts = [left, right, edge] if self.use_edge_features else [left, right]
        return self.lay_concat(parts, **kwargs)

    def get_config(self):
        config = super().get_config()
        config.update({"use_edge_features": self.use_edge_features, "use_bias": self.use_bias, "units": self.units,
                       "has_self_loops": self.has_self_loops, "use_final_activation": self.use_final_activation})
        sub = self._config_source().get_config()
        config.update({key: sub[key] for key in _KERNEL_KEYS + ("activation",)})
        return config


def _kernel_args(local_vars):
    return {key: local_vars[key] for key in _KERNEL_KEYS}


class AttentionHeadGAT(_AttentionHead):

    def __init__(self, units, use_edge_features=False, use_final_activation=True, has_self_loops=True,
                 activation="kgcnn>leaky_relu", use_bias=True, kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(units, use_edge_features, use_final_activation, has_self_loops, activation, use_bias,
                         _kernel_args(locals()), **kwargs)

    def _make_logit_layers(self, activation):
        self.lay_alpha = Dense(1, activation=activation, use_bias=False, **self._kernel_args)

    def _config_source(self):
        return self.lay_alpha

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes (batch,[N],F), edges (batch,[M],Fe), edge_index (batch,[M],2)]`` -> ``(batch,[N],units)``."""
        node, edge, edge_index = inputs
        w_n = self.lay_linear_trafo(node, **kwargs)
        wn_in = self.lay_gather_in([w_n, edge_index], **kwargs)
        wn_out = self.lay_gather_out([w_n, edge_index], **kwargs)
        logits = self.lay_alpha(self._pair_features(wn_in, wn_out, edge, **kwargs), **kwargs)   # (batch,[M],1)
        return self._attend(node, wn_out, logits, edge_index, **kwargs)


class AttentionHeadGATV2(_AttentionHead):

    def __init__(self, units, use_edge_features=False, use_final_activation=True, has_self_loops=True,
                 activation="kgcnn>leaky_relu", use_bias=True, kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(units, use_edge_features, use_final_activation, has_self_loops, activation, use_bias,
                         _kernel_args(locals()), **kwargs)

    def _make_logit_layers(self, activation):
        self.lay_alpha_activation = Dense(self.units, activation=activation, use_bias=self.use_bias, **self._kernel_args)
        self.lay_alpha = Dense(1, activation="linear", use_bias=False, **self._kernel_args)

    def _config_source(self):
        return self.lay_alpha_activation

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes, edges, edge_index]`` as for ``AttentionHeadGAT``."""
        node, edge, edge_index = inputs
        w_n = self.lay_linear_trafo(node, **kwargs)
        n_in = self.lay_gather_in([node, edge_index], **kwargs)
        n_out = self.lay_gather_out([node, edge_index], **kwargs)
        wn_out = self.lay_gather_out([w_n, edge_index], **kwargs)
        hidden = self.lay_alpha_activation(self._pair_features(n_in, n_out, edge, **kwargs), **kwargs)
        return self._attend(node, wn_out, self.lay_alpha(hidden, **kwargs), edge_index, **kwargs)


class MultiHeadGATV2Layer(AttentionHeadGATV2):
    r"""``num_heads`` GATv2 heads in one layer that also hands back the attention logits
    (kgcnn/layers/conv/gat_conv.py:233-323; MEGAN consumes the logits as edge importances).

    Returns ``(h, a)``: node embeddings ``(batch, [N], units * num_heads)`` (``concat_heads``) or their average
    ``(batch, [N], units)``, and the logits of all heads ``(batch, [M], num_heads, 1)``.  Like the reference, every head
    owns three Dense layers (``units`` with the activation, ``units`` with the activation, ``1`` linear) and reuses the
    base class's gathers, concatenation, attention pooling and optional final activation."""

    def __init__(self, units: int, num_heads: int, activation: str = "kgcnn>leaky_relu", use_bias: bool = True,
                 concat_heads: bool = True, **kwargs):
        super().__init__(units=units, activation=activation, use_bias=use_bias, **kwargs)
        self.num_heads = int(num_heads)
        self.concat_heads = concat_heads
        self.head_layers = []
        for _ in range(self.num_heads):
            self.head_layers += [Dense(units, activation=activation, use_bias=use_bias),
                                 Dense(units, activation=activation, use_bias=use_bias),
                                 Dense(1, activation="linear", use_bias=False)]
        self.lay_combine_heads = LazyConcatenate(axis=-1) if concat_heads else LazyAverage()

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes, edges, edge_index]`` -> ``(node embeddings, attention logits)``."""
        node, edge, edge_index = inputs
        n_in = self.lay_gather_in([node, edge_index], **kwargs)
        n_out = self.lay_gather_out([node, edge_index], **kwargs)
        pair = self._pair_features(n_in, n_out, edge, **kwargs)       # identical for every head
        logits, embeddings = [], []
        for k in range(self.num_heads):
            lay_linear, lay_alpha_activation, lay_alpha = self.head_layers[3 * k:3 * k + 3]
            wn_out = self.lay_gather_out([lay_linear(node, **kwargs), edge_index], **kwargs)
            a_ij = lay_alpha(lay_alpha_activation(pair, **kwargs), **kwargs)                     # (batch, [M], 1)
            embeddings.append(self._attend(node, wn_out, a_ij, edge_index, **kwargs))
            logits.append(a_ij)
        # the reference expands every head's logits to (batch, [M], 1, 1) and concatenates on axis -2; with a trailing
        # unit axis that is the last-axis concatenation of the (batch, [M], 1) logits, reshaped
        stacked = self.lay_concat(logits, **kwargs)
        stacked = stacked.with_values(stacked.values.unsqueeze(-1))                              # (batch, [M], K, 1)
        return self.lay_combine_heads(embeddings, **kwargs), stacked

    def get_config(self):
        config = super().get_config()
        config.update({"num_heads": self.num_heads, "concat_heads": self.concat_heads})
        return config
