"""Geometry pre-step layers of SchNet / PaiNN (mirror of the seven hot-path classes of kgcnn/layers/geom.py)."""
import numpy as np
import torch

from .. import _ffi
from ..ops.axis import get_positive_axis
from .base import GraphBaseLayer
from .gather import GatherNodesSelection
from .modules import LazyMultiply, LazySubtract


class NodePosition(GraphBaseLayer):
    r"""Node positions for the two ends of every edge = ``GatherNodesSelection([0, 1])`` (kgcnn/layers/geom.py:14-73)."""

    def __init__(self, selection_index: list = None, **kwargs):
        super().__init__(**kwargs)
        if selection_index is None:
            selection_index = [0, 1]
        self.selection_index = selection_index
        self.layer_gather = GatherNodesSelection(self.selection_index)

    def call(self, inputs, **kwargs):
        return self.layer_gather(inputs, **kwargs)

    def get_config(self):
        config = super().get_config()
        config.update({"selection_index": self.selection_index})
        return config


def _rdc(values, axis_values):
    """View a values tensor as (R, D, C) with D the reduced axis."""
    shape = [int(s) for s in values.shape]
    r = 1
    for s in shape[:axis_values]:
        r *= s
    c = 1
    for s in shape[axis_values + 1:]:
        c *= s
    return r, shape[axis_values], c


class EuclideanNorm(GraphBaseLayer):
    r"""``sqrt(relu(sum_axis x^2))`` with optional eps / inversion (kgcnn/layers/geom.py:127-214)."""

    def __init__(self, axis: int = -1, keepdims: bool = False, invert_norm: bool = False, add_eps: bool = False,
                 no_nan: bool = True, square_norm: bool = False, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis
        self.keepdims = keepdims
        self.invert_norm = invert_norm
        self.square_norm = square_norm
        self.add_eps = add_eps
        self.no_nan = no_nan

    def build(self, input_shape):
        super().build(input_shape)
        self.axis = get_positive_axis(self.axis, len(input_shape))

    @staticmethod
    def _compute_euclidean_norm(inputs, axis: int = -1, keepdims: bool = False, invert_norm: bool = False,
                                add_eps: bool = False, no_nan: bool = True, square_norm: bool = False):
        _ffi.require_device(inputs)
        x = inputs.contiguous()
        ax = axis if axis >= 0 else axis + x.dim()
        r, d, c = _rdc(x, ax)
        shape = list(x.shape)
        out_shape = shape[:ax] + ([1] if keepdims else []) + shape[ax + 1:]
        out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        flags = (1 if invert_norm else 0) | (2 if add_eps else 0) | (4 if no_nan else 0) | (8 if square_norm else 0)
        from ..autograd import EuclideanNorm as NormFn, needs_grad
        if needs_grad(inputs):
            return NormFn.apply(inputs, r, d, c, flags, tuple(out_shape))
        _ffi.call("mp_euclidean_norm_f32", _ffi.ptr(x), r, d, c, flags, _ffi.ptr(out), _ffi.stream())
        return out

    def call(self, inputs, **kwargs):
        return self.map_values(self._compute_euclidean_norm, inputs, axis=self.axis, keepdims=self.keepdims,
                               invert_norm=self.invert_norm, add_eps=self.add_eps, no_nan=self.no_nan,
                               square_norm=self.square_norm)

    def get_config(self):
        config = super().get_config()
        config.update({"axis": self.axis, "keepdims": self.keepdims, "invert_norm": self.invert_norm,
                       "add_eps": self.add_eps, "no_nan": self.no_nan, "square_norm": self.square_norm})
        return config


class ScalarProduct(GraphBaseLayer):
    r"""``sum_axis a*b`` (kgcnn/layers/geom.py:218-281)."""

    def __init__(self, axis=-1, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis

    def build(self, input_shape):
        super().build(input_shape)
        axis = get_positive_axis(self.axis, len(input_shape[0]))
        axis2 = get_positive_axis(self.axis, len(input_shape[1]))
        assert axis2 == axis, "Axis parameter must match on the two input vectors for scalar product."
        self.axis = axis

    @staticmethod
    def _scalar_product(inputs: list, axis: int, **kwargs):
        from ..autograd import ScalarProduct as ProdFn, needs_grad
        if needs_grad(inputs[0], inputs[1]):
            return ProdFn.apply(inputs[0], inputs[1], axis)
        a, b = inputs[0].contiguous(), inputs[1].contiguous()
        _ffi.require_device(a, b)
        r, d, c = _rdc(a, axis)
        shape = list(a.shape)
        out = torch.empty(shape[:axis] + shape[axis + 1:], dtype=torch.float32, device=a.device)
        _ffi.call("mp_scalar_product_f32", _ffi.ptr(a), _ffi.ptr(b), r, d, c, _ffi.ptr(out), _ffi.stream())
        return out

    def call(self, inputs, **kwargs):
        return self.map_values(self._scalar_product, inputs, axis=self.axis)

    def get_config(self):
        config = super().get_config()
        config.update({"axis": self.axis})
        return config


class NodeDistanceEuclidean(GraphBaseLayer):
    r"""``||x_1 - x_2||`` with kept last axis, shape ``(batch, [M], 1)`` (kgcnn/layers/geom.py:285-327)."""

    def __init__(self, add_eps: bool = False, no_nan: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.layer_subtract = LazySubtract()
        self.layer_euclidean_norm = EuclideanNorm(axis=2, keepdims=True, add_eps=add_eps, no_nan=no_nan)

    def call(self, inputs, **kwargs):
        diff = self.layer_subtract(inputs)
        return self.layer_euclidean_norm(diff)

    def get_config(self):
        config = super().get_config()
        conf_norm = self.layer_euclidean_norm.get_config()
        config.update({"add_eps": conf_norm["add_eps"], "no_nan": conf_norm["no_nan"]})
        return config


class EdgeDirectionNormalized(GraphBaseLayer):
    r"""``(r_i - r_j) / ||r_i - r_j||`` with ``divide_no_nan`` (kgcnn/layers/geom.py:331-378)."""

    def __init__(self, add_eps: bool = False, no_nan: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.layer_subtract = LazySubtract()
        self.layer_euclidean_norm = EuclideanNorm(axis=2, keepdims=True, invert_norm=True, add_eps=add_eps,
                                                  no_nan=no_nan)
        self.layer_multiply = LazyMultiply()

    def call(self, inputs, **kwargs):
        diff = self.layer_subtract(inputs)
        norm = self.layer_euclidean_norm(diff)
        return self.layer_multiply([diff, norm])

    def get_config(self):
        config = super().get_config()
        conf_norm = self.layer_euclidean_norm.get_config()
        config.update({"add_eps": conf_norm["add_eps"], "no_nan": conf_norm["no_nan"]})
        return config


class GaussBasisLayer(GraphBaseLayer):
    r"""Gaussian radial basis ``exp(-gamma (d - offset - mu_k)^2)``, ``mu_k = k / bins * distance``,
    ``gamma = 1 / (2 sigma^2)`` (kgcnn/layers/geom.py:514-592)."""

    def __init__(self, bins: int = 20, distance: float = 4.0, sigma: float = 0.4, offset: float = 0.0, **kwargs):
        super().__init__(**kwargs)
        self.bins = int(bins)
        self.distance = float(distance)
        self.offset = float(offset)
        self.sigma = float(sigma)
        self.gamma = 1 / sigma / sigma / 2

    def _compute_gauss_basis(self, inputs):
        _ffi.require_device(inputs)
        d = inputs.contiguous()
        if int(d.shape[-1]) != 1:
            raise ValueError("GaussBasisLayer expects distances of shape (batch, [K], 1)")
        from ..autograd import GaussBasis as GaussFn, needs_grad
        if needs_grad(inputs):
            return GaussFn.apply(inputs, self.bins, self.distance, self.sigma, self.offset)
        m = d.numel()
        out = torch.empty(tuple(d.shape[:-1]) + (self.bins,), dtype=torch.float32, device=d.device)
        _ffi.call("mp_gauss_basis_f32", _ffi.ptr(d), m, self.bins, self.distance, self.sigma, self.offset,
                  _ffi.ptr(out), _ffi.stream())
        return out

    def call(self, inputs, **kwargs):
        return self.map_values(self._compute_gauss_basis, inputs)

    def get_config(self):
        config = super().get_config()
        config.update({"bins": self.bins, "distance": self.distance, "offset": self.offset, "sigma": self.sigma})
        return config


class BesselBasisLayer(GraphBaseLayer):
    r"""Bessel radial basis with polynomial envelope (kgcnn/layers/geom.py:717-805): ``env(d/c) sin(f_k d/c)``,
    trainable ``frequencies`` initialised to ``pi * (1..num_radial)``, envelope zero for ``d/c >= 1``."""

    def __init__(self, num_radial: int, cutoff: float, envelope_exponent: int = 5, envelope_type: str = "poly",
                 **kwargs):
        super().__init__(**kwargs)
        self.num_radial = num_radial
        self.cutoff = cutoff
        self.envelope_exponent = envelope_exponent
        self.envelope_type = str(envelope_type)
        if self.envelope_type not in ["poly"]:
            raise ValueError("Unknown envelope type '%s' in `BesselBasisLayer`." % self.envelope_type)
        self.frequencies = self.add_weight(
            "frequencies", (self.num_radial,),
            initializer=lambda shape: np.pi * np.arange(1, shape[0] + 1, dtype=np.float32))

    def expand_bessel_basis(self, inputs):
        _ffi.require_device(inputs)
        d = inputs.contiguous()
        if int(d.shape[-1]) != 1:
            raise ValueError("BesselBasisLayer expects distances of shape (batch, [K], 1)")
        from ..autograd import BesselBasis as BesselFn, needs_grad
        if needs_grad(inputs):
            return BesselFn.apply(inputs, self.frequencies, self.num_radial, float(self.cutoff),
                                  int(self.envelope_exponent))
        out = torch.empty(tuple(d.shape[:-1]) + (self.num_radial,), dtype=torch.float32, device=d.device)
        _ffi.call("mp_bessel_basis_f32", _ffi.ptr(d), d.numel(), _ffi.ptr(self.frequencies), self.num_radial,
                  float(self.cutoff), int(self.envelope_exponent), _ffi.ptr(out), _ffi.stream())
        return out

    def call(self, inputs, **kwargs):
        return self.map_values(self.expand_bessel_basis, inputs)

    def get_config(self):
        config = super().get_config()
        config.update({"num_radial": self.num_radial, "cutoff": self.cutoff,
                       "envelope_exponent": self.envelope_exponent, "envelope_type": self.envelope_type})
        return config


class CosCutOffEnvelope(GraphBaseLayer):
    r"""``0.5 (cos(pi d / R_c) + 1)`` on clipped distances; ``cutoff=None`` means 1e8 (kgcnn/layers/geom.py:809-856)."""

    def __init__(self, cutoff, **kwargs):
        super().__init__(**kwargs)
        self.cutoff = float(np.abs(cutoff)) if cutoff is not None else 1e8

    def _compute_cutoff_envelope(self, inputs):
        _ffi.require_device(inputs)
        from ..autograd import CosCutoff as CosFn, needs_grad
        if needs_grad(inputs):
            return CosFn.apply(inputs, float(self.cutoff))
        d = inputs.contiguous()
        out = torch.empty_like(d)
        _ffi.call("mp_cos_cutoff_f32", _ffi.ptr(d), d.numel(), float(self.cutoff), _ffi.ptr(out), _ffi.stream())
        return out

    def call(self, inputs, **kwargs):
        return self.map_values(self._compute_cutoff_envelope, inputs)

    def get_config(self):
        config = super().get_config()
        config.update({"cutoff": self.cutoff})
        return config
