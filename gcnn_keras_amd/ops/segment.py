"""Segment reduce by name and segment softmax (kgcnn/ops/segment.py) on the HIP engine."""
import torch

from .. import _ffi

_OPS = {
    "segment_mean": _ffi.MP_MEAN, "mean": _ffi.MP_MEAN, "reduce_mean": _ffi.MP_MEAN,
    "segment_sum": _ffi.MP_SUM, "sum": _ffi.MP_SUM, "reduce_sum": _ffi.MP_SUM,
    "segment_max": _ffi.MP_MAX, "max": _ffi.MP_MAX, "reduce_max": _ffi.MP_MAX,
    "segment_min": _ffi.MP_MIN, "min": _ffi.MP_MIN, "reduce_min": _ffi.MP_MIN,
}


def reduce_op_code(segment_name):
    """Name -> engine op; unknown names raise ``TypeError`` like kgcnn/ops/segment.py:51."""
    if segment_name not in _OPS:
        raise TypeError("Unknown segment operation, choose: 'segment_mean', 'segment_sum', ...")
    return _OPS[segment_name]


def _flat2d(data):
    rows = int(data.shape[0])
    elems = 1
    for d in data.shape[1:]:
        elems *= int(d)
    return data.contiguous().view(rows, max(elems, 1)), rows, max(elems, 1)


def csr_from_sorted_ids(segment_ids, num_segments):
    seg32 = segment_ids.to(torch.int32).contiguous()
    ptr = torch.empty(num_segments + 1, dtype=torch.int32, device=segment_ids.device)
    _ffi.call("mp_csr_from_sorted_i32", _ffi.ptr(seg32) if seg32.numel() else None, int(seg32.numel()),
              num_segments, _ffi.ptr(ptr), _ffi.stream())
    return ptr


def segment_reduce_csr(op, data, ptr, perm, n_out, weight=None, normalize_by_weight=False, seg_ids=None):
    """out[n] = op over rows [ptr[n], ptr[n+1]) of ``data`` (through ``perm`` if given); rows without members are 0.
    ``seg_ids`` (segment id per row, original order) is only needed when a gradient w.r.t. ``data`` is requested."""
    from ..autograd import SegmentSum, needs_grad
    if needs_grad(data):
        if seg_ids is None or normalize_by_weight:
            raise NotImplementedError("gradient needs the segment ids and no weight normalisation")
        return SegmentSum.apply(data, op, ptr, perm, n_out, weight, seg_ids)
    return _segment_reduce_raw(op, data, ptr, perm, n_out, weight, normalize_by_weight)


def _segment_reduce_raw(op, data, ptr, perm, n_out, weight=None, normalize_by_weight=False):
    _ffi.require_device(data, ptr)
    flat, m, elems = _flat2d(data)
    out = torch.empty((n_out,) + tuple(data.shape[1:]), dtype=torch.float32, device=data.device)
    w = None if weight is None else weight.contiguous().view(-1)
    if w is not None and w.numel() != m:
        raise ValueError("weights must hold one value per row (shape (M, 1)), got %s" % (tuple(weight.shape),))
    _ffi.call("mp_segment_reduce_csr_f32", op, _ffi.ptr(flat), m, elems, _ffi.ptr(ptr), _ffi.ptr(perm), n_out,
              _ffi.ptr(w), 1 if normalize_by_weight else 0, _ffi.ptr(out), _ffi.stream())
    return out


def segment_ops_by_name(segment_name: str, data, segment_ids):
    """kgcnn/ops/segment.py:28-52 for SORTED ids: output has ``last id + 1`` rows, missing ids give 0."""
    op = reduce_op_code(segment_name)
    _ffi.require_device(data, segment_ids)
    n_out = int(segment_ids[-1].item()) + 1 if segment_ids.numel() > 0 else 0
    ptr = csr_from_sorted_ids(segment_ids, n_out)
    return segment_reduce_csr(op, data, ptr, None, n_out)


def segment_softmax_csr(data, ptr, perm, n_seg):
    flat, m, elems = _flat2d(data)
    out = torch.empty_like(flat)
    _ffi.call("mp_segment_softmax_csr_f32", _ffi.ptr(flat), m, elems, _ffi.ptr(ptr), _ffi.ptr(perm), n_seg,
              _ffi.ptr(out), _ffi.stream())
    return out.view(data.shape)


def segment_softmax(data, segment_ids, normalize: bool = True):
    """kgcnn/ops/segment.py:5-24 (``normalize=False`` skips the max subtraction; the result is the same function)."""
    _ffi.require_device(data, segment_ids)
    n_seg = int(segment_ids[-1].item()) + 1 if segment_ids.numel() > 0 else 0
    ptr = csr_from_sorted_ids(segment_ids, n_seg)
    return segment_softmax_csr(data, ptr, None, n_seg)
