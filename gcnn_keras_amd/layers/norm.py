"""Normalisation over the feature axis of a ragged tensor's values (mirror of kgcnn/layers/norm.py:8-110).

``GraphLayerNormalization`` is Keras ``LayerNormalization`` applied to ``.values``: the reference maps the user's axis
(counted on the ragged shape ``(batch, [N], F...)``, so it must not be 0) to ``axis - 1`` on the values
(norm.py:45-57).  The engine kernel normalises the last axis, which is the only use in the conv layers
(``GraphSageNodeLayer``, sage_conv.py:65); other axes are rejected.
``GraphBatchNormalization`` (batch statistics + moving averages, a training-time construct) stays out of scope.
"""
import torch

from .. import _ffi
from ..autograd import needs_grad
from ..ragged import RaggedTensor
from .base import GraphBaseLayer


class GraphLayerNormalization(GraphBaseLayer):

    def __init__(self, axis=-1, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros",
                 gamma_initializer="ones", beta_regularizer=None, gamma_regularizer=None, beta_constraint=None,
                 gamma_constraint=None, **kwargs):
        super().__init__(**kwargs)
        if isinstance(axis, (list, tuple)):
            if len(axis) != 1:
                raise NotImplementedError("the engine normalises one axis (the last)")
            axis = axis[0]
        if not isinstance(axis, int):
            raise TypeError("Expected an int or a list/tuple of ints for the argument 'axis', but received: %r" % axis)
        if axis == 0:
            raise ValueError("Positive axis for graph normalization must be > 0 or negative.")
        self.axis = axis
        self.epsilon, self.center, self.scale = float(epsilon), bool(center), bool(scale)
        self.beta_initializer, self.gamma_initializer = beta_initializer, gamma_initializer
        self.beta_regularizer, self.gamma_regularizer = beta_regularizer, gamma_regularizer
        self.beta_constraint, self.gamma_constraint = beta_constraint, gamma_constraint
        self.gamma = self.beta = None

    def build(self, input_shape):
        super().build(input_shape)
        n_dims = len(input_shape)
        axis = self.axis if self.axis >= 0 else n_dims + self.axis
        if axis < 1:
            raise ValueError("The (positive) axis must be > 0.")
        if axis != n_dims - 1:
            raise NotImplementedError("the engine normalises the last axis only")
        self.axis = axis                       # positive after build, like the reference (norm.py:89-90)
        width = int(input_shape[-1])
        if self.scale:                         # Keras creates gamma before beta
            self.gamma = self.add_weight("gamma", (width,), self.gamma_initializer)
        if self.center:
            self.beta = self.add_weight("beta", (width,), self.beta_initializer)

    def call(self, inputs, **kwargs):
        values = inputs.values if isinstance(inputs, RaggedTensor) else inputs
        _ffi.require_device(values)
        if needs_grad(values, self.gamma, self.beta):
            raise NotImplementedError("GraphLayerNormalization has no reverse pass on the engine yet")
        x = values.contiguous()
        width = int(x.shape[-1])
        rows = x.numel() // max(width, 1)
        out = torch.empty_like(x)
        _ffi.call("mp_layer_norm_f32", _ffi.ptr(x), rows, width, _ffi.ptr(self.gamma), _ffi.ptr(self.beta),
                  self.epsilon, _ffi.ptr(out), _ffi.stream())
        return inputs.with_values(out) if isinstance(inputs, RaggedTensor) else out

    def get_config(self):
        config = super().get_config()
        config.update({"axis": self.axis, "epsilon": self.epsilon, "center": self.center, "scale": self.scale,
                       "beta_initializer": self.beta_initializer, "gamma_initializer": self.gamma_initializer,
                       "beta_regularizer": self.beta_regularizer, "gamma_regularizer": self.gamma_regularizer,
                       "beta_constraint": self.beta_constraint, "gamma_constraint": self.gamma_constraint})
        return config
