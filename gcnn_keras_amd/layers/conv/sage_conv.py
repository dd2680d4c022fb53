"""GraphSAGE node / edge updates (mirror of kgcnn/layers/conv/sage_conv.py:11-197) on the engine's primitives.

Node layer (sage_conv.py:83-100): ``n' = LN( MLP_self([ n || pool_j MLP_nb([n_j (|| e_ij)]) ]) )``.
Edge layer (sage_conv.py:179-185): ``e' = LN( MLP([ e || n_i || n_j ]) )``.
The LSTM aggregator of the reference wraps a Keras LSTM and stays out of scope.
"""
from ..base import GraphBaseLayer
from ..gather import GatherNodes, GatherNodesOutgoing
from ..mlp import GraphMLP
from ..modules import LazyConcatenate
from ..norm import GraphLayerNormalization
from ..pooling import PoolingLocalMessages

_MLP_KEYS = ("kernel_regularizer", "activity_regularizer", "bias_regularizer", "kernel_constraint", "bias_constraint",
             "kernel_initializer", "bias_initializer", "use_bias")


def _mlp_args(scope):
    return {key: scope[key] for key in _MLP_KEYS}


def _with_mlp_config(config, mlp):
    conf = mlp.get_config()
    config.update({key: conf[key] for key in _MLP_KEYS + ("activation",)})
    return config


class GraphSageNodeLayer(GraphBaseLayer):

    def __init__(self, units, use_edge_features=False, pooling_method="sum", activation="relu", use_bias=True,
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        if pooling_method in ("LSTM", "lstm"):
            raise NotImplementedError("the LSTM aggregator wraps a Keras LSTM and is out of scope on this engine")
        self.units, self.pooling_method, self.use_edge_features = units, pooling_method, use_edge_features
        mlp_args = _mlp_args(locals())
        self.gather_nodes_outgoing = GatherNodesOutgoing()
        self.concatenate = LazyConcatenate()
        self.update_node_from_neighbors_mlp = GraphMLP(units=units, activation=activation, **mlp_args)
        self.update_node_from_self_mlp = GraphMLP(units=units, activation=activation, **mlp_args)
        self.pooling = PoolingLocalMessages(pooling_method=pooling_method)
        self.normalize_nodes = GraphLayerNormalization(axis=-1)

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes, edge_index]`` or, with ``use_edge_features``, ``[nodes, edges, edge_index]``."""
        if self.use_edge_features:
            node, edge, edge_index = inputs
        else:
            (node, edge_index), edge = inputs, None
        message = self.gather_nodes_outgoing([node, edge_index], **kwargs)
        if edge is not None:
            message = self.concatenate([message, edge], **kwargs)
        message = self.update_node_from_neighbors_mlp(message, **kwargs)
        pooled = self.pooling([node, message, edge_index], **kwargs)
        updated = self.update_node_from_self_mlp(self.concatenate([node, pooled], **kwargs), **kwargs)
        return self.normalize_nodes(updated, **kwargs)

    def get_config(self):
        config = super().get_config()
        config.update({"pooling_method": self.pooling_method, "units": self.units,
                       "use_edge_features": self.use_edge_features})
        return _with_mlp_config(config, self.update_node_from_neighbors_mlp)


class GraphSageEdgeUpdateLayer(GraphBaseLayer):

    def __init__(self, units, activation="relu", use_bias=True, kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", use_normalization=True, **kwargs):
        super().__init__(**kwargs)
        self.units, self.use_normalization = units, use_normalization
        self.gather_nodes = GatherNodes()
        self.concatenate = LazyConcatenate()
        self.update_edge_mlp = GraphMLP(units=units, activation=activation, **_mlp_args(locals()))
        self.normalize_edges = GraphLayerNormalization(axis=-1)

    def call(self, inputs, **kwargs):
        """inputs: ``[nodes, edges, edge_index]`` -> updated edges ``(batch, [M], units)``."""
        node, edge, edge_index = inputs
        pair = self.gather_nodes([node, edge_index], **kwargs)
        edge = self.update_edge_mlp(self.concatenate([edge, pair], **kwargs), **kwargs)
        return self.normalize_edges(edge, **kwargs) if self.use_normalization else edge

    def get_config(self):
        config = super().get_config()
        config.update({"units": self.units, "use_normalization": self.use_normalization})
        return _with_mlp_config(config, self.update_edge_mlp)
