"""SchNet continuous-filter convolution and interaction block (mirror of kgcnn/layers/conv/schnet_conv.py).

``call`` runs the reference's op sequence layer by layer through the engine primitives; the fused single-kernel
version of the same arithmetic lives in ``gcnn_keras_amd.engine`` (csrc/mp_cfconv.hip) and is what
``kgcnn.literature.Schnet`` models use when their configuration allows it.
"""
from ..base import GraphBaseLayer
from ..gather import GatherNodesOutgoing
from ..modules import Dense, LazyAdd, LazyMultiply
from ..pooling import PoolingLocalEdges


class SchNetCFconv(GraphBaseLayer):
    r"""Continuous filter convolution (kgcnn/layers/conv/schnet_conv.py:9-89): two Dense layers on the edge basis,
    multiplied onto the sender's node features, pooled at the receiver."""

    def __init__(self, units, cfconv_pool="segment_sum", use_bias=True, activation="kgcnn>shifted_softplus",
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        self.cfconv_pool = cfconv_pool
        self.units = units
        self.use_bias = use_bias
        kernel_args = {"kernel_regularizer": kernel_regularizer, "activity_regularizer": activity_regularizer,
                       "bias_regularizer": bias_regularizer, "kernel_constraint": kernel_constraint,
                       "bias_constraint": bias_constraint, "kernel_initializer": kernel_initializer,
                       "bias_initializer": bias_initializer}
        self.lay_dense1 = Dense(units=self.units, activation=activation, use_bias=self.use_bias, **kernel_args)
        self.lay_dense2 = Dense(units=self.units, activation="linear", use_bias=self.use_bias, **kernel_args)
        self.lay_sum = PoolingLocalEdges(pooling_method=cfconv_pool)
        self.gather_n = GatherNodesOutgoing()
        self.lay_mult = LazyMultiply()

    def build(self, input_shape):
        super().build(input_shape)
        edge_shape = tuple(input_shape[1])
        self.lay_dense1.ensure_built(edge_shape)
        self.lay_dense2.ensure_built(edge_shape[:-1] + (self.units,))

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes (batch,[N],F), edges (batch,[M],B), edge_index (batch,[M],2)]`` -> ``(batch,[N],F)``."""
        node, edge, indexlist = inputs
        x = self.lay_dense1(edge, **kwargs)
        x = self.lay_dense2(x, **kwargs)
        node2exp = self.gather_n([node, indexlist], **kwargs)
        x = self.lay_mult([node2exp, x], **kwargs)
        x = self.lay_sum([node, x, indexlist], **kwargs)
        return x

    def get_config(self):
        config = super().get_config()
        config.update({"cfconv_pool": self.cfconv_pool, "units": self.units})
        config_dense = self.lay_dense1.get_config()
        for x in ["kernel_regularizer", "activity_regularizer", "bias_regularizer", "kernel_constraint",
                  "bias_constraint", "kernel_initializer", "bias_initializer", "activation", "use_bias"]:
            config.update({x: config_dense[x]})
        return config


class SchNetInteraction(GraphBaseLayer):
    r"""SchNet interaction block (kgcnn/layers/conv/schnet_conv.py:93-174):
    ``n + Dense(lin)(Dense(act)(cfconv(Dense_nobias(n), rbf, idx)))``."""

    def __init__(self, units=128, cfconv_pool="sum", use_bias=True, activation="kgcnn>shifted_softplus",
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        self.cfconv_pool = cfconv_pool
        self.use_bias = use_bias
        self.units = units
        kernel_args = {"kernel_regularizer": kernel_regularizer, "activity_regularizer": activity_regularizer,
                       "bias_regularizer": bias_regularizer, "kernel_constraint": kernel_constraint,
                       "bias_constraint": bias_constraint, "kernel_initializer": kernel_initializer,
                       "bias_initializer": bias_initializer}
        conv_args = {"units": self.units, "use_bias": use_bias, "activation": activation, "cfconv_pool": cfconv_pool}
        self.lay_cfconv = SchNetCFconv(**conv_args, **kernel_args)
        self.lay_dense1 = Dense(units=self.units, activation="linear", use_bias=False, **kernel_args)
        self.lay_dense2 = Dense(units=self.units, activation=activation, use_bias=self.use_bias, **kernel_args)
        self.lay_dense3 = Dense(units=self.units, activation="linear", use_bias=self.use_bias, **kernel_args)
        self.lay_add = LazyAdd()

    def build(self, input_shape):
        super().build(input_shape)
        node_shape = tuple(input_shape[0])
        hidden = node_shape[:-1] + (self.units,)
        self.lay_cfconv.ensure_built([hidden, tuple(input_shape[1]), tuple(input_shape[2])])
        self.lay_dense1.ensure_built(node_shape)
        self.lay_dense2.ensure_built(hidden)
        self.lay_dense3.ensure_built(hidden)

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes, edges, tensor_index]`` -> updated nodes ``(batch,[N],F)``."""
        node, edge, indexlist = inputs
        x = self.lay_dense1(node, **kwargs)
        x = self.lay_cfconv([x, edge, indexlist], **kwargs)
        x = self.lay_dense2(x, **kwargs)
        x = self.lay_dense3(x, **kwargs)
        out = self.lay_add([node, x], **kwargs)
        return out

    def get_config(self):
        config = super().get_config()
        config.update({"cfconv_pool": self.cfconv_pool, "units": self.units, "use_bias": self.use_bias})
        conf_dense = self.lay_dense2.get_config()
        for x in ["activation", "kernel_regularizer", "bias_regularizer", "activity_regularizer",
                  "kernel_constraint", "bias_constraint", "kernel_initializer", "bias_initializer"]:
            config.update({x: conf_dense[x]})
        return config
