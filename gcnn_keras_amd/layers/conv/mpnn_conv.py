"""NMPN message and update layers (mirror of kgcnn/layers/conv/mpnn_conv.py:9-210; Gilmer et al. 2017).

``TrafoEdgeNetMessages`` turns edge features into one ``(F', F)`` matrix per edge (a Dense layer + a reshape view),
``MatMulMessages`` multiplies every edge's matrix with its gathered node row (``mp_batched_matvec_f32``: memory bound on
the matrices, read once with 16-B loads), ``GRUUpdate`` is one Keras GRUCell step (``reset_after=True``) on the flat
node values: two GEMMs on the matrix cores + ``mp_gru_combine_f32``.
"""
import numpy as np
import torch

from ... import _ffi
from ..base import GraphBaseLayer
from ..modules import Dense, _dense_raw

_DENSE_KEYS = ("kernel_regularizer", "activity_regularizer", "bias_regularizer", "kernel_constraint",
               "bias_constraint", "kernel_initializer", "bias_initializer", "activation", "use_bias")


class TrafoEdgeNetMessages(GraphBaseLayer):
    """(batch, [M], F) -> (batch, [M], target_shape[0], target_shape[1]) (mpnn_conv.py:9-65)."""

    def __init__(self, target_shape, activation="linear", use_bias=True, kernel_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        self.target_shape = target_shape
        self._units_out, self._units_in = int(target_shape[0]), int(target_shape[1])
        self.lay_dense = Dense(units=self._units_out * self._units_in, activation=activation, use_bias=use_bias,
                               kernel_regularizer=kernel_regularizer, bias_regularizer=bias_regularizer,
                               activity_regularizer=activity_regularizer, kernel_constraint=kernel_constraint,
                               bias_constraint=bias_constraint, kernel_initializer=kernel_initializer,
                               bias_initializer=bias_initializer)

    def build(self, input_shape):
        super().build(input_shape)
        self.lay_dense.ensure_built(tuple(input_shape))

    def call(self, inputs, **kwargs):
        inputs = self.assert_ragged_input_rank(inputs)
        up = self.lay_dense(inputs, **kwargs)
        return up.with_values(up.values.reshape(int(up.values.shape[0]), self._units_out, self._units_in))

    def get_config(self):
        config = super().get_config()
        config.update({"target_shape": self.target_shape})
        dense = self.lay_dense.get_config()
        config.update({key: dense[key] for key in _DENSE_KEYS})
        return config


class MatMulMessages(GraphBaseLayer):
    """``batch_dot(trafo_mat (batch,[M],F',F), edges (batch,[M],F)) -> (batch,[M],F')`` (mpnn_conv.py:69-108)."""

    def call(self, inputs, **kwargs):
        inputs = self.assert_ragged_input_rank(inputs)
        mat, vec = inputs[0].values.contiguous(), inputs[1].values.contiguous()
        _ffi.require_device(mat, vec)
        if mat.dim() != 3 or vec.dim() != 2 or mat.shape[0] != vec.shape[0] or mat.shape[2] != vec.shape[1]:
            raise ValueError("MatMulMessages expects (M,F',F) matrices and (M,F) messages, got %s and %s"
                             % (tuple(mat.shape), tuple(vec.shape)))
        m, ro, c = (int(d) for d in mat.shape)
        out = torch.empty((m, ro), dtype=torch.float32, device=mat.device)
        _ffi.call("mp_batched_matvec_f32", _ffi.ptr(mat), _ffi.ptr(vec), m, ro, c, _ffi.ptr(out), _ffi.stream())
        return inputs[1].with_values(out)


class GRUUpdate(GraphBaseLayer):
    """``GRUCell(units)(updates, state=nodes)`` on the flat values (mpnn_conv.py:111-210), Keras weight layout:
    kernel ``(in, 3u)``, recurrent_kernel ``(u, 3u)``, bias ``(2, 3u)`` (input / recurrent row), gates ``[z | r | h]``."""

    def __init__(self, units, activation="tanh", recurrent_activation="sigmoid", use_bias=True,
                 kernel_initializer="glorot_uniform", recurrent_initializer="orthogonal", bias_initializer="zeros",
                 kernel_regularizer=None, recurrent_regularizer=None, bias_regularizer=None, kernel_constraint=None,
                 recurrent_constraint=None, bias_constraint=None, dropout=0.0, recurrent_dropout=0.0,
                 reset_after=True, **kwargs):
        super().__init__(**kwargs)
        if not reset_after or dropout or recurrent_dropout:
            raise NotImplementedError("GRUUpdate is built for the Keras default cell: reset_after=True, no dropout")
        self.units = int(units)
        self._conf = {"units": units, "activation": activation, "recurrent_activation": recurrent_activation,
                      "use_bias": use_bias, "kernel_initializer": kernel_initializer,
                      "recurrent_initializer": recurrent_initializer, "bias_initializer": bias_initializer,
                      "kernel_regularizer": kernel_regularizer, "recurrent_regularizer": recurrent_regularizer,
                      "bias_regularizer": bias_regularizer, "kernel_constraint": kernel_constraint,
                      "recurrent_constraint": recurrent_constraint, "bias_constraint": bias_constraint,
                      "dropout": dropout, "recurrent_dropout": recurrent_dropout, "reset_after": reset_after}
        self._act = _ffi.activation_code(activation)
        self._rec = _ffi.activation_code(recurrent_activation)
        self.use_bias = use_bias
        self.kernel = self.recurrent_kernel = self.bias = None

    def build(self, input_shape):
        super().build(input_shape)
        in_dim, u = int(input_shape[1][-1]), self.units
        self.kernel = self.add_weight("gru_cell/kernel", (in_dim, 3 * u), self._conf["kernel_initializer"])
        self.recurrent_kernel = self.add_weight("gru_cell/recurrent_kernel", (u, 3 * u),
                                                self._conf["recurrent_initializer"])
        if self.use_bias:
            self.bias = self.add_weight("gru_cell/bias", (2, 3 * u), self._conf["bias_initializer"])

    def call(self, inputs, **kwargs):
        inputs = self.assert_ragged_input_rank(inputs)
        h, x = inputs[0].values.contiguous(), inputs[1].values.contiguous()
        _ffi.require_device(h, x)
        b_in = self.bias[0].contiguous() if self.bias is not None else None
        b_rec = self.bias[1].contiguous() if self.bias is not None else None
        mx = _dense_raw(x, self.kernel, b_in, 0, 0.0)
        mh = _dense_raw(h, self.recurrent_kernel, b_rec, 0, 0.0)
        out = torch.empty_like(h)
        _ffi.call("mp_gru_combine_f32", _ffi.ptr(mx), _ffi.ptr(mh), _ffi.ptr(h), int(h.shape[0]), self.units, self._act,
                  self._rec, _ffi.ptr(out), _ffi.stream())
        return inputs[0].with_values(out)

    def get_config(self):
        config = super().get_config()
        config.update(self._conf)
        return config
