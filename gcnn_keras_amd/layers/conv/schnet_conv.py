"""SchNet continuous-filter convolution and interaction block on the HIP engine.

API mirror of ``kgcnn.layers.conv.schnet_conv`` (reference kgcnn/layers/conv/schnet_conv.py:9-174): ``SchNetCFconv``
and ``SchNetInteraction`` keep the reference's constructor arguments, input lists and ``get_config`` keys.  Two
execution routes produce the same numbers (tests/test_gpu_fused.py):

* fused: ``SchNetCFconv.call`` hands the whole block - filter MLP, sender gather, product, segment-sum - to ONE kernel
  (``mp_cfconv_fused_f32``, csrc/mp_cfconv.hip) whenever the configuration fits it (128 units, sum pooling, shifted
  softplus, basis width <= 32, no gradient requested);
* layered: otherwise the reference's op sequence (schnet_conv.py:73-79) runs primitive by primitive.
"""
import torch

from ... import _ffi
from ...autograd import needs_grad
from ..base import GraphBaseLayer
from ..gather import GatherNodesOutgoing
from ..modules import Dense, LazyAdd, LazyMultiply
from ..pooling import PoolingLocalEdges

_DENSE_KEYS = ("kernel_regularizer", "activity_regularizer", "bias_regularizer", "kernel_constraint",
               "bias_constraint", "kernel_initializer", "bias_initializer")
_SUM_NAMES = ("sum", "segment_sum", "reduce_sum")
_SSP_NAMES = ("kgcnn>shifted_softplus", "shifted_softplus")


def _pick(local_vars, keys=_DENSE_KEYS):
    return {k: local_vars[k] for k in keys}


def _mirror_dense_config(config, dense_layer, extra):
    """The reference copies these entries of a sub-Dense into the block's own config (schnet_conv.py:85-88, :170-173)."""
    dense_conf = dense_layer.get_config()
    for key in _DENSE_KEYS + tuple(extra):
        config[key] = dense_conf[key]
    return config


class SchNetCFconv(GraphBaseLayer):
    r"""``out_i = pool_{e: recv(e)=i} x_{send(e)} * Dense(units)(Dense(units, act)(edge_e))``
    (reference kgcnn/layers/conv/schnet_conv.py:9-89)."""

    def __init__(self, units, cfconv_pool="segment_sum", use_bias=True, activation="kgcnn>shifted_softplus",
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        dense_args = _pick(locals())
        self.units, self.cfconv_pool, self.use_bias = units, cfconv_pool, use_bias
        self.lay_dense1 = Dense(units=units, activation=activation, use_bias=use_bias, **dense_args)
        self.lay_dense2 = Dense(units=units, activation="linear", use_bias=use_bias, **dense_args)
        self.lay_sum = PoolingLocalEdges(pooling_method=cfconv_pool)
        self.gather_n = GatherNodesOutgoing()
        self.lay_mult = LazyMultiply()
        self._packed = None  # LDS image of the filter weights for the fused kernel (rebuilt when weights change)
        self._packed_key = None

    def build(self, input_shape):
        super().build(input_shape)
        edge_shape = tuple(input_shape[1])
        self.lay_dense1.ensure_built(edge_shape)
        self.lay_dense2.ensure_built(edge_shape[:-1] + (self.units,))

    # -- fused route ----------------------------------------------------------------------------------------------
    def _fused_applicable(self, node, edge):
        return (self.units == 128 and self.cfconv_pool in _SUM_NAMES and self.lay_dense1.activation in _SSP_NAMES
                and node.values.dim() == 2 and int(node.values.shape[1]) == 128 and edge.values.dim() == 2
                and int(edge.values.shape[1]) <= 32 and self.lay_sum.pooling_index == 0
                and self.lay_sum.has_unconnected and not needs_grad(node.values, edge.values))

    def _packed_weights(self, basis):
        tensors = (self.lay_dense1.kernel, self.lay_dense1.bias, self.lay_dense2.kernel, self.lay_dense2.bias)
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors)
        if self._packed is None or key != self._packed_key:
            buf = torch.empty(_ffi.lib().mp_cfconv_packed_floats(), dtype=torch.float32, device=tensors[0].device)
            _ffi.call("mp_cfconv_pack_f32", _ffi.ptr(tensors[0]), _ffi.ptr(tensors[1]), basis, _ffi.ptr(tensors[2]),
                      _ffi.ptr(tensors[3]), _ffi.ptr(buf), _ffi.stream())
            self._packed, self._packed_key = buf, key
        return self._packed

    def _call_fused(self, node, edge, indexlist):
        plan = indexlist.index_plan(node)
        if self.ragged_validate:
            plan.validate()
        _, perm, recv_sorted = plan.csr(0, assume_sorted=self.lay_sum.is_sorted)
        basis = int(edge.values.shape[1])
        out = torch.zeros((plan.N, 128), dtype=torch.float32, device=node.values.device)
        _ffi.call("mp_cfconv_fused_f32", _ffi.ptr(node.values.contiguous()), plan.N, _ffi.ptr(edge.values.contiguous()),
                  basis, _ffi.ptr(self._packed_weights(basis)), _ffi.ptr(recv_sorted.contiguous()),
                  _ffi.ptr(plan.col(1).contiguous()), _ffi.ptr(perm), plan.M, 1, _ffi.ptr(out), _ffi.stream())
        return node.with_values(out)

    # -- layered route: the reference's sequence, kgcnn/layers/conv/schnet_conv.py:73-79 ------------------------------
    def _call_layers(self, node, edge, indexlist, **kwargs):
        filt = self.lay_dense2(self.lay_dense1(edge, **kwargs), **kwargs)
        sender_rows = self.gather_n([node, indexlist], **kwargs)
        messages = self.lay_mult([sender_rows, filt], **kwargs)
        return self.lay_sum([node, messages, indexlist], **kwargs)

    def call(self, inputs, fused=None, **kwargs):
        r"""inputs: ``[nodes (batch,[N],F), edges (batch,[M],B), edge_index (batch,[M],2)]`` -> ``(batch,[N],F)``.
        ``fused``: force (True) / forbid (False) the single-kernel route; default picks it when applicable."""
        node, edge, indexlist = inputs
        can_fuse = self._fused_applicable(node, edge)
        if fused is True and not can_fuse:
            raise ValueError("this SchNetCFconv configuration / input does not fit the fused kernel")
        if can_fuse and fused is not False:
            return self._call_fused(node, edge, indexlist)
        return self._call_layers(node, edge, indexlist, **kwargs)

    def get_config(self):
        config = super().get_config()
        config.update({"cfconv_pool": self.cfconv_pool, "units": self.units})
        return _mirror_dense_config(config, self.lay_dense1, ("activation", "use_bias"))


class SchNetInteraction(GraphBaseLayer):
    r"""``n + Dense(lin)(Dense(act)(cfconv(Dense_nobias(n), edges, idx)))``
    (reference kgcnn/layers/conv/schnet_conv.py:93-174)."""

    def __init__(self, units=128, cfconv_pool="sum", use_bias=True, activation="kgcnn>shifted_softplus",
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        dense_args = _pick(locals())
        self.units, self.cfconv_pool, self.use_bias = units, cfconv_pool, use_bias
        self.lay_cfconv = SchNetCFconv(units=units, use_bias=use_bias, activation=activation, cfconv_pool=cfconv_pool,
                                       **dense_args)
        self.lay_dense1 = Dense(units=units, activation="linear", use_bias=False, **dense_args)
        self.lay_dense2 = Dense(units=units, activation=activation, use_bias=use_bias, **dense_args)
        self.lay_dense3 = Dense(units=units, activation="linear", use_bias=use_bias, **dense_args)
        self.lay_add = LazyAdd()

    def build(self, input_shape):
        super().build(input_shape)
        node_shape = tuple(input_shape[0])
        hidden = node_shape[:-1] + (self.units,)
        self.lay_cfconv.ensure_built([hidden, tuple(input_shape[1]), tuple(input_shape[2])])
        self.lay_dense1.ensure_built(node_shape)
        for lay in (self.lay_dense2, self.lay_dense3):
            lay.ensure_built(hidden)

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes, edges, tensor_index]`` -> updated nodes ``(batch,[N],F)`` (schnet_conv.py:159-165)."""
        node, edge, indexlist = inputs
        update = self.lay_cfconv([self.lay_dense1(node, **kwargs), edge, indexlist], **kwargs)
        nv, uv = node.values, update.values
        if (self.units == 128 and self.lay_dense2.activation in _SSP_NAMES and nv.dim() == 2 and uv.dim() == 2
                and tuple(nv.shape) == tuple(uv.shape) and int(nv.shape[1]) == 128 and nv.dtype == torch.float32
                and not needs_grad(nv, uv)):
            # node side in ONE kernel (csrc/mp_schnet_node.hip, 16-node tiles, both 128x128 GEMMs on FP32 MFMA with the
            # hidden tile handed over in LDS, residual add in the epilogue) instead of Dense, Dense, LazyAdd
            _ffi.require_device(nv, uv)
            out = torch.empty_like(nv)
            _ffi.call("mp_schnet_node_residual_f32", _ffi.ptr(uv.contiguous()), int(nv.shape[0]),
                      _ffi.ptr(self.lay_dense2.kernel), _ffi.ptr(self.lay_dense2.bias),
                      _ffi.ptr(self.lay_dense3.kernel), _ffi.ptr(self.lay_dense3.bias), _ffi.ptr(nv.contiguous()),
                      _ffi.ptr(out), 0, _ffi.stream())
            return node.with_values(out)
        update = self.lay_dense3(self.lay_dense2(update, **kwargs), **kwargs)
        return self.lay_add([node, update], **kwargs)

    def get_config(self):
        config = super().get_config()
        config.update({"cfconv_pool": self.cfconv_pool, "units": self.units, "use_bias": self.use_bias})
        return _mirror_dense_config(config, self.lay_dense2, ("activation",))
