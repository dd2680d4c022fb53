"""Activations of the hot path (kgcnn/ops/activ.py:6-15 ``shifted_softplus``; Keras strings of SchNet/PaiNN/GCN)."""
import torch

from .. import _ffi


def apply_activation(name, x, alpha=0.05):
    code = _ffi.activation_code(name)
    if code == 0:
        return x
    _ffi.require_device(x)
    from ..autograd import Activation, needs_grad
    if needs_grad(x):
        return Activation.apply(x, code, float(alpha))
    xc = x.contiguous()
    out = torch.empty_like(xc)
    _ffi.call("mp_activation_f32", code, float(alpha), _ffi.ptr(xc), xc.numel(), _ffi.ptr(out), _ffi.stream())
    return out


def shifted_softplus(x):
    """``softplus(x) - log(2)`` (kgcnn/ops/activ.py:15)."""
    return apply_activation("kgcnn>shifted_softplus", x)


def softmax(x):
    """Keras ``softmax`` on the last axis."""
    _ffi.require_device(x)
    xc = x.contiguous()
    c = int(xc.shape[-1])
    out = torch.empty_like(xc)
    _ffi.call("mp_softmax_rows_f32", _ffi.ptr(xc), xc.numel() // max(c, 1), c, _ffi.ptr(out), _ffi.stream())
    return out
