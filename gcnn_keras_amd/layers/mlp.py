"""Multilayer perceptron on (ragged) tensors (mirror of kgcnn/layers/mlp.py:12-325; ``MLP`` = ``GraphMLP``).

Per layer: Dense(linear) -> [dropout] -> [normalisation] -> Activation (mlp.py:309-315).  Layer normalisation
(``normalization_technique`` "graph_layer" / "GraphLayerNormalization", and Keras' "layer" / "LayerNormalization", which is
the same arithmetic on the values' last axis) runs on the engine (``mp_layer_norm_f32``); dropout is the identity in a
forward with ``training`` false (Keras semantics) and raises when training; the batch-statistics techniques ("batch",
"graph_batch") are training-time constructs outside the hot path and raise.
"""
from .base import GraphBaseLayer
from .modules import Activation, Dense
from .norm import GraphLayerNormalization


class MLPBase(GraphBaseLayer):
    r"""Argument broadcasting of kgcnn/layers/mlp.py:12-242: every per-layer argument may be a single value or a
    list matching ``units``."""

    _KEYS = ["activation", "activity_regularizer", "units", "use_bias", "kernel_regularizer", "bias_regularizer",
             "kernel_initializer", "bias_initializer", "kernel_constraint", "bias_constraint",
             "use_dropout", "use_normalization", "normalization_technique", "rate", "noise_shape", "seed",
             "axis", "momentum", "epsilon", "center", "scale"]

    def __init__(self, units, use_bias=True, activation=None, activity_regularizer=None, kernel_regularizer=None,
                 bias_regularizer=None, kernel_initializer="glorot_uniform", bias_initializer="zeros",
                 kernel_constraint=None, bias_constraint=None, use_normalization=False,
                 normalization_technique="batch", axis=-1, momentum=0.99, epsilon=0.001, center=True, scale=True,
                 use_dropout=False, rate=None, noise_shape=None, seed=None, **kwargs):
        super().__init__(**kwargs)
        local_kw = dict(locals())
        if isinstance(units, int):
            units = [units]
        if not isinstance(units, list):
            raise ValueError("Units must be a list or a single int for `MLP`.")
        local_kw["units"] = units
        self._depth = len(units)
        for key in self._KEYS:
            value = local_kw[key]
            if not isinstance(value, (list, tuple)):
                value = [value for _ in range(self._depth)]
            if len(value) != self._depth:
                raise ValueError("Provide matching list of units %s and %s or simply a single value." % (units, key))
            setattr(self, "_conf_" + key, list(value))

    def get_config(self):
        config = super().get_config()
        for key in self._KEYS:
            config.update({key: getattr(self, "_conf_" + key)})
        return config


class MLP(MLPBase):
    r"""Stack of Dense layers with activations (kgcnn/layers/mlp.py:246-325)."""

    def __init__(self, units, **kwargs):
        super().__init__(units=units, **kwargs)
        self.mlp_dense_layer_list = [
            Dense(units=self._conf_units[i], use_bias=self._conf_use_bias[i], activation="linear",
                  kernel_initializer=self._conf_kernel_initializer[i], bias_initializer=self._conf_bias_initializer[i],
                  name=self.name + "_dense_" + str(i)) for i in range(self._depth)]
        self.mlp_activation_layer_list = [
            Activation(activation=self._conf_activation[i], name=self.name + "_act_" + str(i))
            for i in range(self._depth)]

        # created after the Dense and Activation lists, like the reference (mlp.py:258-293): fixes the weight order
        self.mlp_norm_layer_list = [None] * self._depth
        for i in range(self._depth):
            if not self._conf_use_normalization[i]:
                continue
            technique = self._conf_normalization_technique[i]
            if technique in ("graph_layer", "GraphLayerNormalization", "layer", "LayerNormalization"):   # mlp.py:285-290
                self.mlp_norm_layer_list[i] = GraphLayerNormalization(
                    axis=self._conf_axis[i], epsilon=self._conf_epsilon[i], center=self._conf_center[i],
                    scale=self._conf_scale[i], name=self.name + "_norm_" + str(i))
            elif technique in ("batch", "BatchNormalization", "graph_batch", "GraphBatchNormalization"):
                raise NotImplementedError("batch normalisation inside MLP is a training-time construct outside the "
                                          "forward hot path")
            else:
                raise NotImplementedError("Normalization via %s not supported." % technique)          # mlp.py:291-293

    def build(self, input_shape):
        super().build(input_shape)
        shape = tuple(input_shape)
        for i in range(self._depth):
            self.mlp_dense_layer_list[i].ensure_built(shape)
            shape = shape[:-1] + (self._conf_units[i],)
            if self.mlp_norm_layer_list[i] is not None:
                self.mlp_norm_layer_list[i].ensure_built(shape)

    def call(self, inputs, **kwargs):
        x = inputs
        if any(self._conf_use_dropout) and kwargs.get("training"):
            raise NotImplementedError("dropout in a training forward is outside the hot path")
        for i in range(self._depth):
            if self.mlp_norm_layer_list[i] is not None:   # Dense -> norm -> Activation (mlp.py:309-315)
                x = self.mlp_dense_layer_list[i](x, **kwargs)
                x = self.mlp_norm_layer_list[i](x, **kwargs)
                x = self.mlp_activation_layer_list[i](x, **kwargs)
                continue
            # Dense(linear) followed by Activation is computed as one GEMM with the activation in its epilogue;
            # same arithmetic, one pass less.
            d = self.mlp_dense_layer_list[i]
            act = self.mlp_activation_layer_list[i].activation
            saved = d.activation
            d.activation = act
            try:
                x = d(x, **kwargs)
            finally:
                d.activation = saved
        return x


GraphMLP = MLP
