"""tensor_scatter_nd by name (kgcnn/ops/scatter.py) for the (node, relation) scatter of RelationalPoolingLocalEdges."""
import torch

from .. import _ffi

_OPS = {"segment_sum": _ffi.MP_SUM, "sum": _ffi.MP_SUM, "reduce_sum": _ffi.MP_SUM, "add": _ffi.MP_SUM,
        "segment_max": _ffi.MP_MAX, "max": _ffi.MP_MAX, "reduce_max": _ffi.MP_MAX,
        "segment_min": _ffi.MP_MIN, "min": _ffi.MP_MIN, "reduce_min": _ffi.MP_MIN}


def scatter_op_code(segment_name):
    if segment_name not in _OPS:
        raise TypeError("Unknown pooling, choose: 'mean', 'sum', ...")
    return _OPS[segment_name]


def tensor_scatter_nd_ops_by_name(segment_name, tensor, indices, updates, name=None):
    """kgcnn/ops/scatter.py:5-26 for ``tensor`` (N, R, F...), ``indices`` (M, 2), ``updates`` (M, F...)."""
    op = scatter_op_code(segment_name)
    _ffi.require_device(tensor, indices, updates)
    out = tensor.clone().contiguous()
    n, r = int(out.shape[0]), int(out.shape[1])
    m = int(updates.shape[0])
    elems = 1
    for d in updates.shape[1:]:
        elems *= int(d)
    recv = indices[:, 0].to(torch.int32).contiguous()
    rel = indices[:, 1].to(torch.int32).contiguous()
    _ffi.call("mp_scatter_relational_f32", op, _ffi.ptr(updates.contiguous()), m, max(elems, 1), _ffi.ptr(recv),
              _ffi.ptr(rel), n, r, _ffi.ptr(out), _ffi.stream())
    return out
