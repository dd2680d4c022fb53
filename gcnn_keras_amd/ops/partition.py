"""Row-partition conversion and the sample<->batch index shift (kgcnn/ops/partition.py).

``partition_row_indexing`` is the a1 row of SURVEY.md section 8: the shift itself runs in the HIP kernel
``mp_shift_index_i64``; the tiny partition-type conversions around it are host-side torch glue on (G,)-sized
int64 tensors (the reference's cumsum / pad / repeat on the same tensors).
"""
import torch

from .. import _ffi

_LEN = ["row_length", "row_lengths"]
_SPL = ["row_split", "row_splits"]
_STA = ["row_start", "row_starts"]
_LIM = ["row_limit", "row_limits"]


def _pad_front(x):
    return torch.cat([torch.zeros(1, dtype=x.dtype, device=x.device), x])


def _seg_count(ids):
    if ids.numel() == 0:
        return torch.zeros(0, dtype=ids.dtype, device=ids.device)
    return torch.bincount(ids, minlength=int(ids[-1].item()) + 1).to(ids.dtype)


def change_partition_by_name(in_partition, in_partition_type: str, out_partition_type: str):
    """Same table as kgcnn/ops/partition.py:5-93 (1-D partition tensors, RaggedTensor naming)."""
    p = in_partition
    if in_partition_type == out_partition_type:
        return p
    if in_partition_type in _LEN and out_partition_type in _SPL:
        return _pad_front(torch.cumsum(p, 0))
    if in_partition_type in _LEN and out_partition_type == "value_rowids":
        return torch.repeat_interleave(torch.arange(p.shape[0], device=p.device, dtype=torch.int32), p)
    if in_partition_type in _LEN and out_partition_type in _STA:
        return torch.cumsum(p, 0) - p
    if in_partition_type in _LEN and out_partition_type in _LIM:
        return torch.cumsum(p, 0)
    if in_partition_type in _SPL and out_partition_type in _LEN:
        return p[1:] - p[:-1]
    if in_partition_type in _SPL and out_partition_type == "value_rowids":
        part_sum = p[1:] - p[:-1]
        return torch.repeat_interleave(torch.arange(part_sum.shape[0], device=p.device, dtype=torch.int32), part_sum)
    if in_partition_type in _SPL and out_partition_type in _LIM:
        return p[1:]
    if in_partition_type in _SPL and out_partition_type in _STA:
        return p[:-1]
    if in_partition_type == "value_rowids" and out_partition_type in _LEN:
        return _seg_count(p)
    if in_partition_type == "value_rowids" and out_partition_type in _SPL:
        return _pad_front(torch.cumsum(_seg_count(p), 0))
    if in_partition_type == "value_rowids" and out_partition_type in _LIM:
        return torch.cumsum(_seg_count(p), 0)
    if in_partition_type == "value_rowids" and out_partition_type in _STA:
        c = _seg_count(p)
        return torch.cumsum(c, 0) - c
    if in_partition_type in _STA:
        raise ValueError("Can not infer partition scheme from row_starts alone, missing nvals")
    if in_partition_type in _LIM and out_partition_type in _LEN:
        s = _pad_front(p)
        return s[1:] - s[:-1]
    if in_partition_type in _LIM and out_partition_type == "value_rowids":
        s = _pad_front(p)
        part_sum = s[1:] - s[:-1]
        return torch.repeat_interleave(torch.arange(part_sum.shape[0], device=p.device, dtype=torch.int32), part_sum)
    if in_partition_type in _LIM and out_partition_type in _SPL:
        return _pad_front(p)
    if in_partition_type in _LIM and out_partition_type in _STA:
        return _pad_front(p)[:-1]
    raise TypeError("Unknown partition scheme, use: 'value_rowids', 'row_splits', 'row_lengths', etc.")


def partition_row_indexing(tensor_index, part_target, part_index, partition_type_target, partition_type_index,
                           from_indexing: str = "sample", to_indexing: str = "batch"):
    """kgcnn/ops/partition.py:97-162: shift an index tensor between per-sample and disjoint-batch indexing."""
    if to_indexing == from_indexing:
        return tensor_index
    if to_indexing == "batch" and from_indexing == "sample":
        direction = 1
    elif to_indexing == "sample" and from_indexing == "batch":
        direction = -1
    else:
        raise TypeError("ERROR:kgcnn: Unknown index change, use: 'sample', 'batch', ...")
    _ffi.require_device(tensor_index, part_target, part_index)
    nod_splits = change_partition_by_name(part_target, partition_type_target, "row_splits").to(torch.int64).contiguous()
    edge_splits = change_partition_by_name(part_index, partition_type_index, "row_splits").to(torch.int64).contiguous()
    in_dtype = tensor_index.dtype
    idx = tensor_index.to(torch.int64).contiguous()
    m = int(idx.shape[0])
    k = 1
    for d in idx.shape[1:]:
        k *= int(d)
    out = torch.empty_like(idx)
    _ffi.call("mp_shift_index_i64", _ffi.ptr(idx), m, max(k, 1), _ffi.ptr(nod_splits), _ffi.ptr(edge_splits),
              int(edge_splits.shape[0]) - 1, direction, _ffi.ptr(out), _ffi.stream())
    return out.to(in_dtype)
