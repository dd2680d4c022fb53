"""Layer protocol and ``GraphBaseLayer`` (mirrors kgcnn/layers/base.py:8-171 without Keras).

A layer is called as ``layer(inputs, **kwargs)`` with ``inputs`` a (list of) :class:`RaggedTensor` of ragged rank 1
(kgcnn/layers/base.py:88-92), builds its weights on first call, and round-trips its constructor arguments through
``get_config()`` with the reference's keys.  Weights are torch tensors resident in HBM.
"""
import itertools

import numpy as np
import torch

from ..ragged import RaggedTensor

_name_counters = {}
_seed_counter = itertools.count(1)
_global_seed = [0]


def set_global_seed(seed):
    """Seed for weight initialisers (Keras ``glorot_uniform`` / ``uniform`` draw from here in creation order)."""
    global _seed_counter
    _global_seed[0] = int(seed)
    _seed_counter = itertools.count(1)


def _next_rng():
    return np.random.default_rng([_global_seed[0], next(_seed_counter)])


def _camel_to_snake(name):
    out = []
    for i, ch in enumerate(name):
        if ch.isupper() and i > 0 and not name[i - 1].isupper():
            out.append("_")
        out.append(ch.lower())
    return "".join(out)


_weight_epoch = [0]


def weight_epoch():
    """Counts weight-tensor OBJECT changes anywhere (``add_weight``, ``layer.kernel = other_tensor``): the fused routes
    compare it on every call, so a replaced tensor is seen at once.  In-place updates are seen through the tensors' version
    counters; only ``tensor.data = ...`` (same object, same version, other storage) waits for a route's periodic full walk."""
    return _weight_epoch[0]


class Layer:
    """Just enough of ``ks.layers.Layer``: naming, lazy build, weights, ``get_config``."""

    def __setattr__(self, key, value):
        if torch.is_tensor(value):
            old = self.__dict__.get(key)
            if torch.is_tensor(old) and old is not value:
                ws = self.__dict__.get("_weights")
                if ws is not None:            # the attribute names a weight: keep the weight list on the new object
                    for i, (n, t) in enumerate(ws):
                        if t is old:
                            ws[i] = (n, value)
            _weight_epoch[0] += 1
        object.__setattr__(self, key, value)

    def __init__(self, name=None, trainable=True, dtype="float32", **kwargs):
        if kwargs:
            raise TypeError("Unknown layer arguments %s" % sorted(kwargs))
        if name is None:
            base = _camel_to_snake(type(self).__name__)
            n = _name_counters.get(base, 0)
            _name_counters[base] = n + 1
            name = base if n == 0 else "%s_%d" % (base, n)
        self.name = name
        self.trainable = trainable
        self.dtype = dtype
        self.built = False
        self._weights = []  # (name, tensor) in creation order

    # -- weights -----------------------------------------------------------------------------------------------
    def add_weight(self, name, shape, initializer="glorot_uniform", device=None, fan=None):
        if device is None:  # weights live in HBM; without a GPU they are only constructible, not usable
            device = "cuda" if torch.cuda.is_available() else "cpu"
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        rng = _next_rng()
        if callable(initializer):
            arr = np.asarray(initializer(shape), dtype=np.float32)
        elif initializer in ("zeros", "Zeros"):
            arr = np.zeros(shape, dtype=np.float32)
        elif initializer in ("ones", "Ones"):
            arr = np.ones(shape, dtype=np.float32)
        elif initializer in ("glorot_uniform", "GlorotUniform"):
            fan_in, fan_out = fan if fan is not None else (shape[0], shape[-1])
            limit = np.sqrt(6.0 / (fan_in + fan_out))
            arr = rng.uniform(-limit, limit, size=shape).astype(np.float32)
        elif initializer in ("orthogonal", "Orthogonal"):
            # Keras Orthogonal: QR of a normal matrix (sign-fixed), rows >= cols handled by transposition
            rows, cols = int(np.prod(shape[:-1])), shape[-1]
            q, r = np.linalg.qr(rng.normal(size=(max(rows, cols), min(rows, cols))))
            q = q * np.sign(np.diag(r))
            arr = (q if rows >= cols else q.T).reshape(shape).astype(np.float32)
        elif initializer in ("uniform", "RandomUniform", "random_uniform"):
            arr = rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
        else:
            raise ValueError("Unsupported initializer %r" % (initializer,))
        t = torch.from_numpy(np.array(arr, dtype=np.float32, order="C")).to(device)   # C order, 0-d stays 0-d
        self._weights.append((name, t))
        _weight_epoch[0] += 1
        return t

    def sublayers(self):
        out = []
        for v in self.__dict__.values():
            if isinstance(v, Layer):
                out.append(v)
            elif isinstance(v, (list, tuple)):
                out.extend(x for x in v if isinstance(x, Layer))
        return out

    @property
    def weights(self):
        out = list(self._weights)
        for sub in self.sublayers():
            out.extend((sub.name + "/" + n, t) for n, t in sub.weights)
        return out

    def get_weights(self):
        return [t.detach().cpu().numpy() for _, t in self.weights]

    def set_weights(self, arrays):
        ws = self.weights
        if len(ws) != len(arrays):
            raise ValueError("Layer %s expects %d weight arrays, got %d" % (self.name, len(ws), len(arrays)))
        for (n, t), a in zip(ws, arrays):
            a = np.asarray(a, dtype=np.float32)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError("Shape mismatch for %s: %s vs %s" % (n, tuple(a.shape), tuple(t.shape)))
            t.copy_(torch.from_numpy(a))

    # -- call protocol -----------------------------------------------------------------------------------------
    @staticmethod
    def _shape_of(x):
        if isinstance(x, (list, tuple)):
            return [Layer._shape_of(v) for v in x]
        if isinstance(x, RaggedTensor):
            return x.shape
        return tuple(x.shape)

    def build(self, input_shape):
        self.built = True

    def call(self, inputs, **kwargs):
        raise NotImplementedError

    def ensure_built(self, input_shape):
        """Create the weights for ``input_shape`` (last axis must be static) if that has not happened yet."""
        if not self.built:
            self.build(input_shape)
            self.built = True
        return self

    def __call__(self, inputs, **kwargs):
        self.ensure_built(self._shape_of(inputs))
        return self.call(inputs, **kwargs)

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable, "dtype": self.dtype}

    @classmethod
    def from_config(cls, config):
        return cls(**config)


class GraphBaseLayer(Layer):
    """Mirror of ``kgcnn.layers.base.GraphBaseLayer`` (kgcnn/layers/base.py:8-171): the four behaviour flags
    ``node_indexing, ragged_validate, is_sorted, has_unconnected`` are part of the drop-in boundary."""

    def __init__(self, node_indexing: str = "sample", ragged_validate: bool = False, is_sorted: bool = False,
                 has_unconnected: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.node_indexing = node_indexing
        self.ragged_validate = ragged_validate
        self.is_sorted = is_sorted
        self.has_unconnected = has_unconnected
        self._supports_ragged_inputs = True
        self._kgcnn_info = {"node_indexing": self.node_indexing, "ragged_validate": self.ragged_validate,
                            "is_sorted": self.is_sorted, "has_unconnected": self.has_unconnected}
        if self.node_indexing != "sample":
            raise ValueError("Indexing for disjoint representation is not supported as of version 1.0")
        self._add_layer_config_to_self = {}

    def get_config(self):
        config = super().get_config()
        config.update({"node_indexing": self.node_indexing, "ragged_validate": self.ragged_validate,
                       "is_sorted": self.is_sorted, "has_unconnected": self.has_unconnected})
        for key, value in self._add_layer_config_to_self.items():
            if hasattr(self, key) and getattr(self, key) is not None:
                layer_conf = getattr(self, key).get_config()
                for x in value:
                    if x in layer_conf:
                        config.update({x: layer_conf[x]})
        return config

    def assert_ragged_input_rank(self, inputs, mask=None, ragged_rank: int = 1):
        """kgcnn/layers/base.py:70-110: ragged inputs pass, dense (batch, N, F...) tensors are cast to ragged."""
        if mask is not None:
            raise ValueError("Using `mask` argument in `assert_ragged_input_rank` is not yet supported.")

        def validate_or_cast(x):
            if isinstance(x, RaggedTensor):
                if ragged_rank is not None:
                    assert x.ragged_rank == ragged_rank, "'%s' must have input with ragged_rank=%s." % (
                        self.name, ragged_rank)
                return x
            elif isinstance(x, torch.Tensor):
                if ragged_rank is None:
                    raise ValueError("Casting to ragged without `ragged_rank` information is not supported.")
                if ragged_rank != 1:
                    raise ValueError("Casting to ragged is only supported for ragged_rank=1 at the moment.")
                if x.dim() <= ragged_rank:
                    raise ValueError(
                        "Rank of inputs must be > ragged_rank but found '%s <= %s' " % (x.dim(), ragged_rank))
                b, n = int(x.shape[0]), int(x.shape[1])
                return RaggedTensor.from_row_lengths(x.reshape((b * n,) + tuple(x.shape[2:])).contiguous(),
                                                     torch.full((b,), n, dtype=torch.int64, device=x.device))
            raise ValueError("Unsupported tensor type '%s' in '%s'." % (type(x), self.name))

        if isinstance(inputs, (list, tuple)):
            return [validate_or_cast(x) for x in inputs]
        return validate_or_cast(inputs)

    def map_values(self, fun, inputs, **kwargs):
        """kgcnn/layers/base.py:112-171: call ``fun`` on the ``.values`` of one / a list of ragged tensors and rewrap
        with the first input's partition; ``axis`` arguments > 1 are shifted by one for the values tensor."""
        kwargs_values = dict(kwargs)
        if "axis" in kwargs:
            axis = kwargs["axis"]
            axis_values = None
            kwargs_values = None
            if isinstance(axis, int):
                if axis > 1:
                    axis_values = axis - 1
            elif isinstance(axis, (list, tuple)):
                if all(x > 1 for x in axis):
                    axis_values = [x - 1 for x in axis]
            if axis_values is not None:
                kwargs_values = dict(kwargs)
                kwargs_values["axis"] = axis_values
        if isinstance(inputs, list) and kwargs_values is not None:
            if all(isinstance(x, RaggedTensor) for x in inputs) and not self.ragged_validate:
                out = fun([x.values for x in inputs], **kwargs_values)
                if isinstance(out, list):
                    return [inputs[i].with_values(x) for i, x in enumerate(out)]
                return inputs[0].with_values(out)
        elif isinstance(inputs, RaggedTensor) and kwargs_values is not None:
            out = fun(inputs.values, **kwargs_values)
            if isinstance(out, list):
                return [inputs.with_values(x) for x in out]
            return inputs.with_values(out)
        if isinstance(inputs, RaggedTensor):
            print("WARNING: Layer %s fail call on value Tensor of ragged Tensor." % self.name)
        if isinstance(inputs, list) and any(isinstance(x, RaggedTensor) for x in inputs):
            print("WARNING: Layer %s fail call on value Tensor for ragged Tensor in list." % self.name)
        return fun(inputs, **kwargs)

    call_on_values_tensor_of_ragged = map_values
