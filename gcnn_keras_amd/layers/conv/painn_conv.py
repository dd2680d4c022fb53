"""PaiNN message and update blocks (mirror of kgcnn/layers/conv/painn_conv.py) on the HIP engine primitives.

Scalar channel ``(batch, [N], F)``, equivariant channel ``(batch, [N], 3, F)`` (the reference docstring's ``(F, 3)``
at painn_conv.py:95 is wrong, see SURVEY.md section 8a note 12)."""
import torch

from ...ops.axis import get_positive_axis
from ..base import GraphBaseLayer
from ..gather import GatherNodesOutgoing
from ..geom import EuclideanNorm, ScalarProduct
from ..modules import Dense, ExpandDims, LazyAdd, LazyConcatenate, LazyMultiply, split_last
from ..pooling import PoolingLocalEdges


def _kernel_args(kernel_regularizer, bias_regularizer, activity_regularizer, kernel_constraint, bias_constraint,
                 kernel_initializer, bias_initializer):
    return {"kernel_regularizer": kernel_regularizer, "activity_regularizer": activity_regularizer,
            "bias_regularizer": bias_regularizer, "kernel_constraint": kernel_constraint,
            "bias_constraint": bias_constraint, "kernel_initializer": kernel_initializer,
            "bias_initializer": bias_initializer}


_DENSE_CONFIG_KEYS = ("kernel_regularizer", "activity_regularizer", "bias_regularizer", "kernel_constraint",
                      "bias_constraint", "kernel_initializer", "bias_initializer", "activation", "use_bias")


def _with_dense_config(config, dense_layer):
    """Copy the sub-Dense entries the reference exposes on the block itself (painn_conv.py:117-125, :216-224)."""
    dense_conf = dense_layer.get_config()
    config.update({key: dense_conf[key] for key in _DENSE_CONFIG_KEYS})
    return config


class SplitEmbedding(GraphBaseLayer):
    """Split the last axis of a ragged tensor into equal parts (kgcnn/layers/conv/painn_conv.py:301-346)."""

    def __init__(self, num_or_size_splits, axis=-1, num=None, **kwargs):
        super().__init__(**kwargs)
        self.num_or_size_splits = num_or_size_splits
        self.axis = axis
        self.out_num = num

    def build(self, input_shape):
        super().build(input_shape)
        self.axis = get_positive_axis(self.axis, len(input_shape))
        if self.axis <= 1:
            raise ValueError("Can not split tensor at axis <= 1.")

    def call(self, inputs, **kwargs):
        self.assert_ragged_input_rank(inputs, ragged_rank=1)
        if not isinstance(self.num_or_size_splits, int) or self.axis != inputs.values.dim():
            raise NotImplementedError("SplitEmbedding is built for an equal split of the last axis")
        return [inputs.with_values(x) for x in split_last(inputs.values, self.num_or_size_splits)]

    def get_config(self):
        config = super().get_config()
        config.update({"num_or_size_splits": self.num_or_size_splits, "axis": self.axis, "num": self.out_num})
        return config


class PAiNNconv(GraphBaseLayer):
    """Continuous filter convolution block of PaiNN (kgcnn/layers/conv/painn_conv.py:12-125)."""

    def __init__(self, units, conv_pool="sum", use_bias=True, activation="swish", cutoff=None,
                 kernel_regularizer=None, bias_regularizer=None, activity_regularizer=None, kernel_constraint=None,
                 bias_constraint=None, kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        self.conv_pool = conv_pool
        self.units = units
        self.use_bias = use_bias
        self.cutoff = cutoff
        kernel_args = _kernel_args(kernel_regularizer, bias_regularizer, activity_regularizer, kernel_constraint,
                                   bias_constraint, kernel_initializer, bias_initializer)
        self.lay_dense1 = Dense(units=self.units, activation=activation, use_bias=self.use_bias, **kernel_args)
        self.lay_phi = Dense(units=self.units * 3, activation="linear", use_bias=self.use_bias, **kernel_args)
        self.lay_w = Dense(units=self.units * 3, activation="linear", use_bias=self.use_bias, **kernel_args)
        self.lay_split = SplitEmbedding(3, axis=-1)
        self.lay_sum = PoolingLocalEdges(pooling_method=conv_pool)
        self.lay_sum_v = PoolingLocalEdges(pooling_method=conv_pool)
        self.gather_n = GatherNodesOutgoing()
        self.gather_v = GatherNodesOutgoing()
        self.lay_mult = LazyMultiply()
        if self.cutoff is not None:
            self.lay_mult_cutoff = LazyMultiply()
        self.lay_exp_vv = ExpandDims(axis=-2)
        self.lay_exp_vw = ExpandDims(axis=-2)
        self.lay_exp_r = ExpandDims(axis=-1)
        self.lay_mult_vv = LazyMultiply()
        self.lay_mult_vw = LazyMultiply()
        self.lay_add = LazyAdd()

    def build(self, input_shape):
        super().build(input_shape)
        node_shape, rbf_shape = tuple(input_shape[0]), tuple(input_shape[2])
        self.lay_dense1.ensure_built(node_shape)
        self.lay_phi.ensure_built(node_shape[:-1] + (self.units,))
        self.lay_w.ensure_built(rbf_shape)

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes (b,[N],F), equivariant (b,[N],3,F), rbf (b,[M],B), envelope (b,[M],1), r_ij (b,[M],3),
        edge_index (b,[M],2)]`` -> ``(ds (b,[N],F), dv (b,[N],3,F))``."""
        node, equivariant, rbf, envelope, r_ij, indexlist = inputs
        s = self.lay_dense1(node)
        s = self.lay_phi(s)
        fused = self._fused_message(node, s, equivariant, rbf, envelope, r_ij, indexlist)
        if fused is not None:
            return fused
        s = self.gather_n([s, indexlist])
        w = self.lay_w(rbf)
        if self.cutoff is not None:
            w = self.lay_mult_cutoff([w, envelope])
        sw = self.lay_mult([s, w])
        sw1, sw2, sw3 = self.lay_split(sw)
        ds = self.lay_sum([node, sw1, indexlist])
        vj = self.gather_v([equivariant, indexlist])
        sw2 = self.lay_exp_vv(sw2)
        dv1 = self.lay_mult_vv([sw2, vj])
        sw3 = self.lay_exp_vw(sw3)
        r_ij = self.lay_exp_r(r_ij)
        dv2 = self.lay_mult_vw([sw3, r_ij])
        dv = self.lay_add([dv1, dv2])
        dv = self.lay_sum_v([node, dv, indexlist])
        return ds, dv

    def _fused_message(self, node, s, equivariant, rbf, envelope, r_ij, indexlist):
        """Edge side of the block in one kernel (``mp_painn_message_fused_f32``) when the configuration allows:
        128 units, sum pooling, basis <= 32, no gradient requested (forces use the layer sequence)."""
        from ... import _ffi
        from ...autograd import needs_grad
        if (self.units != 128 or self.conv_pool not in ("sum", "segment_sum", "reduce_sum")
                or int(rbf.values.shape[-1]) > 32 or rbf.values.dim() != 2 or equivariant.values.dim() != 3
                or int(equivariant.values.shape[1]) != 3
                or needs_grad(s.values, equivariant.values, rbf.values, r_ij.values, envelope.values)):
            return None
        plan = indexlist.index_plan(node)
        pool = self.lay_sum
        if pool.pooling_index != 0 or not pool.has_unconnected:
            return None
        ptr, perm, _ = plan.csr(0, assume_sorted=pool.is_sorted)
        sv, vv = s.values.contiguous(), equivariant.values.contiguous()
        ds = torch.empty((plan.N, 128), dtype=torch.float32, device=sv.device)
        dv = torch.empty((plan.N, 3, 128), dtype=torch.float32, device=sv.device)
        env = envelope.values.contiguous().view(-1) if self.cutoff is not None else None
        _ffi.call("mp_painn_message_fused_f32", _ffi.ptr(sv), _ffi.ptr(vv), plan.N, _ffi.ptr(rbf.values.contiguous()),
                  int(rbf.values.shape[-1]), _ffi.ptr(env), _ffi.ptr(r_ij.values.contiguous()),
                  _ffi.ptr(self.lay_w.kernel), _ffi.ptr(self.lay_w.bias), _ffi.ptr(ptr), _ffi.ptr(perm),
                  _ffi.ptr(plan.col(1).contiguous()), plan.M, _ffi.ptr(ds), _ffi.ptr(dv), _ffi.stream())
        return node.with_values(ds), equivariant.with_values(dv)

    def get_config(self):
        config = super().get_config()
        config.update({"conv_pool": self.conv_pool, "units": self.units, "cutoff": self.cutoff})
        return _with_dense_config(config, self.lay_dense1)


class PAiNNUpdate(GraphBaseLayer):
    """Node-local update block of PaiNN (kgcnn/layers/conv/painn_conv.py:129-224)."""

    def __init__(self, units, use_bias=True, activation="swish", kernel_regularizer=None, bias_regularizer=None,
                 activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                 kernel_initializer="glorot_uniform", bias_initializer="zeros", **kwargs):
        super().__init__(**kwargs)
        self.units = units
        self.use_bias = use_bias
        kernel_args = _kernel_args(kernel_regularizer, bias_regularizer, activity_regularizer, kernel_constraint,
                                   bias_constraint, kernel_initializer, bias_initializer)
        self.lay_dense1 = Dense(units=self.units, activation=activation, use_bias=self.use_bias, **kernel_args)
        self.lay_lin_u = Dense(self.units, activation="linear", use_bias=False, **kernel_args)
        self.lay_lin_v = Dense(self.units, activation="linear", use_bias=False, **kernel_args)
        self.lay_a = Dense(units=self.units * 3, activation="linear", use_bias=self.use_bias, **kernel_args)
        self.lay_scalar_prod = ScalarProduct(axis=2)
        self.lay_norm = EuclideanNorm(axis=2)
        self.lay_concat = LazyConcatenate(axis=-1)
        self.lay_split = SplitEmbedding(3, axis=-1)
        self.lay_mult = LazyMultiply()
        self.lay_exp_v = ExpandDims(axis=-2)
        self.lay_mult_vv = LazyMultiply()
        self.lay_add = LazyAdd()

    def build(self, input_shape):
        super().build(input_shape)
        node_shape, equiv_shape = tuple(input_shape[0]), tuple(input_shape[1])
        self.lay_dense1.ensure_built(node_shape[:-1] + (node_shape[-1] + self.units,))
        self.lay_lin_u.ensure_built(equiv_shape)
        self.lay_lin_v.ensure_built(equiv_shape)
        self.lay_a.ensure_built(node_shape[:-1] + (self.units,))

    def call(self, inputs, **kwargs):
        r"""inputs: ``[nodes (b,[N],F), equivariant (b,[N],3,F)]`` -> ``(ds, dv)``."""
        node, equivariant = inputs
        v_v = self.lay_lin_v(equivariant, **kwargs)
        v_u = self.lay_lin_u(equivariant, **kwargs)
        v_prod = self.lay_scalar_prod([v_u, v_v], **kwargs)
        v_norm = self.lay_norm(v_v, **kwargs)
        a = self.lay_concat([node, v_norm], **kwargs)
        a = self.lay_dense1(a, **kwargs)
        a = self.lay_a(a, **kwargs)
        a_vv, a_sv, a_ss = self.lay_split(a, **kwargs)
        a_vv = self.lay_exp_v(a_vv, **kwargs)
        dv = self.lay_mult_vv([a_vv, v_u], **kwargs)
        ds = self.lay_mult([v_prod, a_sv], **kwargs)
        ds = self.lay_add([ds, a_ss], **kwargs)
        return ds, dv

    def get_config(self):
        config = super().get_config()
        config.update({"units": self.units})
        return _with_dense_config(config, self.lay_dense1)


class EquivariantInitialize(GraphBaseLayer):
    """Initial equivariant tensor ``(batch, [N], dim, F)`` (kgcnn/layers/conv/painn_conv.py:228-297)."""

    def __init__(self, dim=3, method: str = "zeros", value: float = 1.0, stddev: float = 1.0, **kwargs):
        super().__init__(**kwargs)
        self.dim = int(dim)
        self.method = str(method)
        self.value = float(value)
        self.stddev = float(stddev)

    def build(self, input_shape):
        super().build(input_shape)
        assert len(input_shape) >= 3, "ERROR:kgcnn: Need input shape of form (batch, None, F_dim)."

    def call(self, inputs, **kwargs):
        inputs = self.assert_ragged_input_rank(inputs)
        v = inputs.values
        shape = (int(v.shape[0]), self.dim) + tuple(int(s) for s in v.shape[1:])
        if self.method == "zeros":
            out = torch.zeros(shape, dtype=v.dtype, device=v.device)
        elif self.method == "eps":
            out = torch.full(shape, 1e-7, dtype=v.dtype, device=v.device)  # ks.backend.epsilon()
        elif self.method == "ones":
            out = torch.ones(shape, dtype=v.dtype, device=v.device)
        elif self.method == "const":
            out = torch.full(shape, self.value, dtype=v.dtype, device=v.device)
        elif self.method == "node":
            out = v.unsqueeze(1).expand(shape).contiguous()
        else:
            raise ValueError("Unknown initialization method %s" % self.method)
        return inputs.with_values(out)

    def get_config(self):
        config = super().get_config()
        config.update({"dim": self.dim, "method": self.method, "value": self.value, "stddev": self.stddev})
        return config
