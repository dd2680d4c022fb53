"""Set2Set readout (mirror of kgcnn/layers/pool/set2set.py:13-305; Vinyals et al. 2016, as used by NMPN and MEGNet).

``T`` rounds of: query ``q`` from one LSTM step on ``q* = [q, r]``, attention of ``q`` over the set's rows
(``e = pool_f(m * q)``, softmax within each graph), read ``r = sum_n a_n m_n``.  As in the reference the LSTM is a
stateless Keras layer called on a length-1 sequence, i.e. every round is one LSTM step FROM THE ZERO STATE: the recurrent
kernel and the forget gate have no effect on the output (they exist as weights, in Keras order, so that ``set_weights``
accepts a reference checkpoint).  Kernels: input GEMM on the matrix cores (``mp_dense_f32``), gate arithmetic
(``mp_lstm_zero_state_f32``), per-graph broadcast / softmax / weighted pooling on the ragged primitives.
"""
import torch

from ... import _ffi
from ...ragged import RaggedTensor
from ..base import GraphBaseLayer
from ..modules import _binary_raw, _concat_last_raw, _dense_raw


class PoolingSet2Set(GraphBaseLayer):

    def __init__(self, channels, T=3, pooling_method="mean", init_qstar="mean", activation="tanh",
                 recurrent_activation="sigmoid", use_bias=True, kernel_initializer="glorot_uniform",
                 recurrent_initializer="orthogonal", bias_initializer="zeros", unit_forget_bias=True,
                 kernel_regularizer=None, recurrent_regularizer=None, bias_regularizer=None, activity_regularizer=None,
                 kernel_constraint=None, recurrent_constraint=None, bias_constraint=None, dropout=0.0,
                 recurrent_dropout=0.0, implementation=2, return_sequences=False, return_state=False, go_backwards=False,
                 stateful=False, time_major=False, unroll=False, **kwargs):
        super().__init__(**kwargs)
        if pooling_method not in ("mean", "sum"):
            raise TypeError("ERROR:kgcnn: Unknown pooling, choose: 'mean', 'sum', ...")
        if stateful or dropout or recurrent_dropout:
            raise NotImplementedError("stateful / dropout LSTM variants are outside the forward hot path")
        self.channels, self.T = int(channels), int(T)
        self.pooling_method, self.init_qstar = pooling_method, init_qstar
        self._lstm_conf = {"activation": activation, "recurrent_activation": recurrent_activation, "use_bias": use_bias,
                           "kernel_initializer": kernel_initializer, "recurrent_initializer": recurrent_initializer,
                           "bias_initializer": bias_initializer, "unit_forget_bias": unit_forget_bias,
                           "kernel_regularizer": kernel_regularizer, "recurrent_regularizer": recurrent_regularizer,
                           "bias_regularizer": bias_regularizer, "activity_regularizer": activity_regularizer,
                           "kernel_constraint": kernel_constraint, "recurrent_constraint": recurrent_constraint,
                           "bias_constraint": bias_constraint, "dropout": dropout,
                           "recurrent_dropout": recurrent_dropout, "implementation": implementation,
                           "return_sequences": return_sequences, "return_state": return_state,
                           "go_backwards": go_backwards, "stateful": stateful, "time_major": time_major,
                           "unroll": unroll}
        self._act = _ffi.activation_code(activation)
        self._rec = _ffi.activation_code(recurrent_activation)
        self.use_bias = use_bias
        self.lstm_kernel = self.lstm_recurrent_kernel = self.lstm_bias = None

    def build(self, input_shape):
        super().build(input_shape)
        u = self.channels
        # Keras LSTM weights: kernel (2u, 4u), recurrent_kernel (u, 4u), bias (4u) with the forget gate at 1
        self.lstm_kernel = self.add_weight("lstm/kernel", (2 * u, 4 * u), self._lstm_conf["kernel_initializer"])
        self.lstm_recurrent_kernel = self.add_weight("lstm/recurrent_kernel", (u, 4 * u),
                                                     self._lstm_conf["recurrent_initializer"])
        if self.use_bias:
            def bias_init(shape):
                import numpy as np
                b = np.zeros(shape, dtype=np.float32)
                if self._lstm_conf["unit_forget_bias"]:
                    b[u:2 * u] = 1.0
                return b
            self.lstm_bias = self.add_weight("lstm/bias", (4 * u,), bias_init)

    # -- pieces -------------------------------------------------------------------------------------------------------
    def _attend(self, m, q, splits, ptr32, n, g):
        """r = sum_n softmax_graph(pool_f(m * q_rep)) m  (set2set.py:191-199)."""
        f = self.channels
        dev = m.device
        qt = torch.empty((n, f), dtype=torch.float32, device=dev)
        _ffi.call("mp_repeat_rows_f32", _ffi.ptr(q), _ffi.ptr(splits), g, f, n, _ffi.ptr(qt), _ffi.stream())
        et = torch.empty((n, 1), dtype=torch.float32, device=dev)
        _ffi.call("mp_scalar_product_f32", _ffi.ptr(m), _ffi.ptr(qt), n, f, 1, _ffi.ptr(et), _ffi.stream())
        if self.pooling_method == "mean":
            et = _binary_raw(_ffi.MP_MUL, et, torch.full((1, 1), 1.0 / f, dtype=torch.float32, device=dev))
        at = torch.empty_like(et)
        _ffi.call("mp_segment_softmax_csr_f32", _ffi.ptr(et), n, 1, _ffi.ptr(ptr32), None, g, _ffi.ptr(at),
                  _ffi.stream())
        rt = torch.empty((g, f), dtype=torch.float32, device=dev)
        _ffi.call("mp_pool_graph_f32", _ffi.MP_SUM, _ffi.ptr(m), _ffi.ptr(splits), g, f, _ffi.ptr(at.view(-1)),
                  _ffi.ptr(rt), _ffi.stream())
        return rt

    def call(self, inputs, **kwargs):
        """inputs: ragged ``(batch, [N], channels)`` -> ``(batch, 1, 2 * channels)``."""
        inputs = self.assert_ragged_input_rank(inputs)
        m = inputs.values.contiguous()
        _ffi.require_device(m)
        if m.dim() != 2 or int(m.shape[1]) != self.channels:
            raise ValueError("PoolingSet2Set(channels=%d) got rows of width %s" % (self.channels, tuple(m.shape[1:])))
        n, g, f = int(m.shape[0]), inputs.nrows(), self.channels
        splits = inputs.row_splits
        ptr32 = splits.to(torch.int32)
        dev = m.device
        if self.init_qstar == "mean":
            q = torch.empty((g, f), dtype=torch.float32, device=dev)
            _ffi.call("mp_pool_graph_f32", _ffi.MP_MEAN, _ffi.ptr(m), _ffi.ptr(splits), g, f, None, _ffi.ptr(q),
                      _ffi.stream())
            qstar = _concat_last_raw([q, self._attend(m, q, splits, ptr32, n, g)])
        else:
            qstar = torch.zeros((g, 2 * f), dtype=torch.float32, device=dev)
        for _ in range(self.T):
            z = _dense_raw(qstar, self.lstm_kernel, self.lstm_bias, 0, 0.0)            # (G, 4u): x W + b, h0 = 0
            q = torch.empty((g, f), dtype=torch.float32, device=dev)
            _ffi.call("mp_lstm_zero_state_f32", _ffi.ptr(z), g, f, self._act, self._rec, _ffi.ptr(q), _ffi.stream())
            qstar = _concat_last_raw([q, self._attend(m, q, splits, ptr32, n, g)])
        return qstar.unsqueeze(1)

    def get_config(self):
        config = super().get_config()
        config.update({"channels": self.channels, "T": self.T, "pooling_method": self.pooling_method,
                       "init_qstar": self.init_qstar})
        config.update(self._lstm_conf)
        return config
