"""Ragged-aware Dense / Activation / Lazy* layers (mirror of kgcnn/layers/modules.py) on the HIP engine."""
import torch

from .. import _ffi
from ..ops.activ import apply_activation, softmax
from ..ops.axis import get_positive_axis
from ..ragged import RaggedTensor
from .base import GraphBaseLayer


def _activation_name(activation):
    if activation is None:
        return "linear"
    if isinstance(activation, dict):
        return activation.get("class_name", activation.get("name"))
    return activation


def _dense_raw(x, kernel, bias, act_code, alpha):
    k = int(x.shape[-1])
    u = int(kernel.shape[1])
    xc = x.contiguous()
    rows = xc.numel() // max(k, 1)
    out = torch.empty(tuple(x.shape[:-1]) + (u,), dtype=torch.float32, device=x.device)
    tiles = -(-rows // 64) * -(-u // 64)
    if tiles <= 128 and k >= 512:
        # few 64x64 output tiles, long contraction (GCN's 1433 input features): cut k over workgroups to fill the chip
        splits = max(1, min(256 // tiles, k // 128, 64))
        ws = torch.empty((splits, max(rows, 1), u), dtype=torch.float32, device=x.device)
        _ffi.call("mp_dense_splitk_f32", _ffi.ptr(xc), rows, k, _ffi.ptr(kernel), _ffi.ptr(bias), u, act_code,
                  float(alpha), splits, _ffi.ptr(ws), ws.numel() * 4, _ffi.ptr(out), _ffi.stream())
        return out
    _ffi.call("mp_dense_f32", _ffi.ptr(xc), rows, k, _ffi.ptr(kernel), _ffi.ptr(bias), u, act_code, float(alpha),
              _ffi.ptr(out), _ffi.stream())
    return out


def dense_values(x, kernel, bias, activation="linear", alpha=0.05):
    """``act(x @ kernel + bias)`` on the last axis of a values tensor via ``mp_dense_f32`` (FP32 MFMA)."""
    from ..autograd import Dense as DenseFn, needs_grad
    _ffi.require_device(x, kernel)
    if x.dtype != torch.float32:
        raise TypeError("Dense expects float32 values, got %s" % x.dtype)
    if int(x.shape[-1]) != int(kernel.shape[0]):
        raise ValueError("Dense kernel expects last dimension %d, got %d" % (int(kernel.shape[0]), int(x.shape[-1])))
    name = _activation_name(activation)
    code = 0 if name == "softmax" else _ffi.activation_code(name)
    if needs_grad(x):
        out = DenseFn.apply(x, kernel, bias, code, float(alpha))
    else:
        out = _dense_raw(x, kernel, bias, code, alpha)
    return softmax(out) if name == "softmax" else out


class DenseEmbedding(GraphBaseLayer):
    r"""Dense layer on the flat values of a ragged tensor, :math:`\sigma(xW + b)` (kgcnn/layers/modules.py:15-90).
    Kernel layout ``(in, units)`` and initialisers as in Keras (``glorot_uniform`` / ``zeros``)."""

    def __init__(self, units: int, activation=None, use_bias: bool = True, kernel_initializer="glorot_uniform",
                 bias_initializer="zeros", kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None,
                 kernel_constraint=None, bias_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.units = int(units)
        self.activation = _activation_name(activation)
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer
        self.kernel_regularizer = kernel_regularizer
        self.bias_regularizer = bias_regularizer
        self.activity_regularizer = activity_regularizer
        self.kernel_constraint = kernel_constraint
        self.bias_constraint = bias_constraint
        self.kernel = None
        self.bias = None

    def build(self, input_shape):
        super().build(input_shape)
        self.kernel = self.add_weight("kernel", (int(input_shape[-1]), self.units), self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", (self.units,), self.bias_initializer)

    def call(self, inputs, **kwargs):
        if isinstance(inputs, RaggedTensor):
            return inputs.with_values(dense_values(inputs.values, self.kernel, self.bias, self.activation))
        return dense_values(inputs, self.kernel, self.bias, self.activation)

    def get_config(self):
        config = super().get_config()
        config.update({"units": self.units, "activation": self.activation, "use_bias": self.use_bias,
                       "kernel_initializer": self.kernel_initializer, "bias_initializer": self.bias_initializer,
                       "kernel_regularizer": self.kernel_regularizer, "bias_regularizer": self.bias_regularizer,
                       "activity_regularizer": self.activity_regularizer,
                       "kernel_constraint": self.kernel_constraint, "bias_constraint": self.bias_constraint})
        return config


Dense = DenseEmbedding


class ActivationEmbedding(GraphBaseLayer):
    """Activation on the values of a ragged tensor (kgcnn/layers/modules.py:94-138)."""

    def __init__(self, activation, activity_regularizer=None, **kwargs):
        super().__init__(**kwargs)
        self.activation = _activation_name(activation)
        self.activity_regularizer = activity_regularizer

    def _apply(self, x):
        if self.activation == "softmax":
            return softmax(x)
        return apply_activation(self.activation, x)

    def call(self, inputs, **kwargs):
        if isinstance(inputs, RaggedTensor):
            return inputs.with_values(self._apply(inputs.values))
        return self._apply(inputs)

    def get_config(self):
        config = super().get_config()
        config.update({"activation": self.activation, "activity_regularizer": self.activity_regularizer})
        return config


Activation = ActivationEmbedding


class DropoutEmbedding(GraphBaseLayer):
    """Dropout (kgcnn/layers/modules.py:142-183): identity outside training - the engine is forward / inference only."""

    def __init__(self, rate, noise_shape=None, seed=None, **kwargs):
        super().__init__(**kwargs)
        self.rate, self.noise_shape, self.seed = rate, noise_shape, seed

    def call(self, inputs, training=False, **kwargs):
        if training:
            raise NotImplementedError("training-mode dropout is outside the forward hot path")
        return inputs

    def get_config(self):
        config = super().get_config()
        config.update({"rate": self.rate, "noise_shape": self.noise_shape, "seed": self.seed})
        return config


Dropout = DropoutEmbedding


def binary_values(op, a, b):
    """Broadcasting elementwise op on two values tensors of equal rank (<= 3 non-unit groups)."""
    from ..autograd import Binary, needs_grad
    if needs_grad(a, b):
        return Binary.apply(a, b, op)
    return _binary_raw(op, a, b)


def _binary_raw(op, a, b):
    _ffi.require_device(a, b)
    if a.dim() != b.dim():
        raise ValueError("operands must have equal rank: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    out_shape = []
    for sa, sb in zip(a.shape, b.shape):
        if sa != sb and 1 not in (sa, sb):
            raise ValueError("shapes %s and %s do not broadcast" % (tuple(a.shape), tuple(b.shape)))
        out_shape.append(max(int(sa), int(sb)))
    # collapse to (R, D1, D2): R = rows, D2 = last axis, D1 = everything in between
    if len(out_shape) == 1:
        dims = [out_shape[0], 1, 1]
        da, db = [int(a.shape[0]), 1, 1], [int(b.shape[0]), 1, 1]
    elif len(out_shape) == 2:
        dims = [out_shape[0], 1, out_shape[1]]
        da, db = [int(a.shape[0]), 1, int(a.shape[1])], [int(b.shape[0]), 1, int(b.shape[1])]
    else:
        mid_a = [int(s) for s in a.shape[1:-1]]
        mid_b = [int(s) for s in b.shape[1:-1]]
        mid_o = out_shape[1:-1]

        def prod(v):
            p = 1
            for x in v:
                p *= x
            return p

        for m in (mid_a, mid_b):
            if prod(m) not in (1, prod(mid_o)):
                raise NotImplementedError("partial broadcast inside the middle axes is not supported")
        dims = [out_shape[0], prod(mid_o), out_shape[-1]]
        da = [int(a.shape[0]), prod(mid_a), int(a.shape[-1])]
        db = [int(b.shape[0]), prod(mid_b), int(b.shape[-1])]

    def strides(d):
        s2 = 0 if (d[2] == 1 and dims[2] != 1) else 1
        s1 = 0 if (d[1] == 1 and dims[1] != 1) else d[2]
        s0 = 0 if (d[0] == 1 and dims[0] != 1) else d[1] * d[2]
        return [s0, s1, s2]

    ac, bc = a.contiguous(), b.contiguous()
    out = torch.empty(out_shape, dtype=torch.float32, device=a.device)
    _ffi.call("mp_binary_f32", op, _ffi.ptr(ac), _ffi.int64_array(strides(da)), _ffi.ptr(bc),
              _ffi.int64_array(strides(db)), dims[0], dims[1], dims[2], _ffi.ptr(out), _ffi.stream())
    return out


def _reduce_list(op, values):
    out = values[0]
    for v in values[1:]:
        out = binary_values(op, out, v)
    return out


class LazyAdd(GraphBaseLayer):
    r"""Sum of a list of (ragged) tensors on their values (kgcnn/layers/modules.py:187-212)."""

    def call(self, inputs, **kwargs):
        return self.map_values(lambda vals: _reduce_list(_ffi.MP_ADD, vals), inputs, **kwargs)


class LazySubtract(GraphBaseLayer):
    r"""``inputs[0] - inputs[1]`` on values (kgcnn/layers/modules.py:216-241)."""

    def call(self, inputs, **kwargs):
        if len(inputs) != 2:
            raise ValueError("A `Subtract` layer should be called on exactly 2 inputs")
        return self.map_values(lambda vals: binary_values(_ffi.MP_SUB, vals[0], vals[1]), inputs, **kwargs)


class LazyAverage(GraphBaseLayer):
    r"""Element-wise average of a list of tensors (kgcnn/layers/modules.py:245-271)."""

    def call(self, inputs, **kwargs):
        def avg(vals):
            s = _reduce_list(_ffi.MP_ADD, vals)
            inv = torch.full((1,) * s.dim(), 1.0 / len(vals), dtype=torch.float32, device=s.device)
            return binary_values(_ffi.MP_MUL, s, inv)
        return self.map_values(avg, inputs, **kwargs)


class LazyMultiply(GraphBaseLayer):
    r"""Element-wise product of a list of tensors, broadcasting unit axes (kgcnn/layers/modules.py:275-301)."""

    def call(self, inputs, **kwargs):
        return self.map_values(lambda vals: _reduce_list(_ffi.MP_MUL, vals), inputs, **kwargs)


def concat_last(values):
    """``tf.concat(values, axis=-1)`` by strided column-block copies."""
    from ..autograd import ConcatLast, needs_grad
    if needs_grad(*values):
        return ConcatLast.apply(*values)
    return _concat_last_raw(values)


def _concat_last_raw(values):
    _ffi.require_device(*values)
    lead = tuple(values[0].shape[:-1])
    widths = [int(v.shape[-1]) for v in values]
    total = sum(widths)
    rows = 1
    for d in lead:
        rows *= int(d)
    out = torch.empty(lead + (total,), dtype=values[0].dtype, device=values[0].device)
    off = 0
    for v, w in zip(values, widths):
        if tuple(v.shape[:-1]) != lead:
            raise ValueError("concat operands differ outside the concat axis")
        _ffi.call("mp_copy_cols_f32", _ffi.ptr(v.contiguous()), w, 0, _ffi.ptr(out), total, off, rows, w, _ffi.stream())
        off += w
    return out


def split_last(value, num):
    """``tf.split(value, num, axis=-1)`` into contiguous tensors."""
    from ..autograd import SplitLast, needs_grad
    if needs_grad(value):
        return list(SplitLast.apply(value, num))
    return _split_last_raw(value, num)


def _split_last_raw(value, num):
    _ffi.require_device(value)
    vc = value.contiguous()
    total = int(vc.shape[-1])
    if total % num:
        raise ValueError("cannot split %d columns into %d equal parts" % (total, num))
    w = total // num
    rows = vc.numel() // max(total, 1)
    outs = []
    for i in range(num):
        o = torch.empty(tuple(vc.shape[:-1]) + (w,), dtype=vc.dtype, device=vc.device)
        _ffi.call("mp_copy_cols_f32", _ffi.ptr(vc), total, i * w, _ffi.ptr(o), w, 0, rows, w, _ffi.stream())
        outs.append(o)
    return outs


class LazyConcatenate(GraphBaseLayer):
    r"""Concatenate a list of tensors along ``axis`` (kgcnn/layers/modules.py:305-364); the engine supports the last
    axis, which is what every hot-path model uses."""

    def __init__(self, axis=-1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def build(self, input_shape):
        super().build(input_shape)
        if not isinstance(input_shape, (tuple, list)) or len(input_shape) < 1:
            raise ValueError("A `Concatenate` layer should be called on a list of at least 1 input. "
                             f"Received: input_shape={input_shape}")
        ranks = set(len(shape) for shape in input_shape)
        if len(ranks) == 1:
            self.axis = get_positive_axis(self.axis, len(input_shape[0]))

    def call(self, inputs, **kwargs):
        def cat(vals, axis):
            if axis != vals[0].dim() - 1:
                raise NotImplementedError("LazyConcatenate is built for the last axis only")
            return concat_last(vals)
        return self.map_values(cat, inputs, axis=self.axis)

    def get_config(self):
        config = super().get_config()
        config.update({"axis": self.axis})
        return config


class ExpandDims(GraphBaseLayer):
    r"""``tf.expand_dims`` on the values (kgcnn/layers/modules.py:368-416); a pure view."""

    def __init__(self, axis: int = -1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def build(self, input_shape):
        super().build(input_shape)
        if len(input_shape) == 0:
            return
        self.axis = get_positive_axis(self.axis, len(input_shape) + 1)

    def call(self, inputs, **kwargs):
        return self.map_values(lambda v, axis: v.unsqueeze(axis), inputs, axis=self.axis)

    def get_config(self):
        config = super().get_config()
        config.update({"axis": self.axis})
        return config


class ZerosLike(GraphBaseLayer):
    r"""Zero tensor with the partition of the input (kgcnn/layers/modules.py:420-446)."""

    def call(self, inputs, **kwargs):
        return self.map_values(torch.zeros_like, inputs)


class OptionalInputEmbedding(GraphBaseLayer):
    r"""Optional ``Embedding`` of integer-valued node numbers (kgcnn/layers/modules.py:450-534); table
    ``(input_dim, output_dim)``, ``uniform`` init; float inputs are cast to int32 like Keras does."""

    def __init__(self, input_dim, output_dim, use_embedding=False, embeddings_initializer="uniform",
                 embeddings_regularizer=None, activity_regularizer=None, embeddings_constraint=None, mask_zero=False,
                 input_length=None, **kwargs):
        super().__init__(**kwargs)
        self.use_embedding = use_embedding
        self.input_dim, self.output_dim = input_dim, output_dim
        self.embeddings_initializer = embeddings_initializer
        self.embeddings_regularizer = embeddings_regularizer
        self.activity_regularizer = activity_regularizer
        self.embeddings_constraint = embeddings_constraint
        self.mask_zero = mask_zero
        self.input_length = input_length
        self.embeddings = None

    def build(self, input_shape):
        super().build(input_shape)
        if self.use_embedding:
            self.embeddings = self.add_weight("embeddings", (self.input_dim, self.output_dim),
                                              self.embeddings_initializer)

    def call(self, inputs, **kwargs):
        if not self.use_embedding:
            return inputs
        vals = inputs.values if isinstance(inputs, RaggedTensor) else inputs
        _ffi.require_device(vals)
        numbers = vals.to(torch.float32).contiguous()
        n = numbers.numel()
        out = torch.empty(tuple(numbers.shape) + (self.output_dim,), dtype=torch.float32, device=vals.device)
        _ffi.call("mp_embedding_f32", _ffi.ptr(self.embeddings), self.input_dim, self.output_dim, _ffi.ptr(numbers), n,
                  _ffi.ptr(out), None, _ffi.stream())
        return inputs.with_values(out) if isinstance(inputs, RaggedTensor) else out

    def get_config(self):
        config = super().get_config()
        config.update({"use_embedding": self.use_embedding})
        if self.use_embedding:
            config.update({"input_dim": self.input_dim, "output_dim": self.output_dim,
                           "embeddings_initializer": self.embeddings_initializer,
                           "embeddings_regularizer": self.embeddings_regularizer,
                           "activity_regularizer": self.activity_regularizer,
                           "embeddings_constraint": self.embeddings_constraint, "mask_zero": self.mask_zero,
                           "input_length": self.input_length})
        return config
