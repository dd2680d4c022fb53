"""Minimal ragged carrier with the ``tf.RaggedTensor`` surface the kgcnn layers use (ragged_rank 1 only).

The reference unwraps every ragged tensor to ``.values`` + ``.row_splits`` immediately
(kgcnn/layers/gather.py:76-83, kgcnn/layers/pooling.py:53-61); this class is exactly that pair, resident in HBM
as torch tensors (values: float32 / int64, row_splits: int64).  It also owns the per-batch *index plan*: the
sample->batch shift that the reference recomputes inside every gather and pooling call
(kgcnn/ops/partition.py:97-162) is computed once per (index tensor, node partition) and cached here.
"""
import numpy as np
import torch

from . import _ffi


class RaggedTensor:
    ragged_rank = 1

    def __init__(self, values, row_splits, validate=False):
        if row_splits.dtype != torch.int64:
            row_splits = row_splits.to(torch.int64)
        self.values = values
        self.row_splits = row_splits
        self._splits_host = None
        self._plans = {}
        if validate:
            s = self.row_splits_host()
            if s[0] != 0 or np.any(np.diff(s) < 0) or s[-1] != values.shape[0]:
                raise ValueError("row_splits do not partition the values")

    # ---- constructors (tf.RaggedTensor.from_*) ------------------------------------------------------------
    @classmethod
    def from_row_splits(cls, values, row_splits, validate=False):
        return cls(values, row_splits, validate=validate)

    @classmethod
    def from_row_lengths(cls, values, row_lengths, validate=False):
        row_lengths = torch.as_tensor(row_lengths, dtype=torch.int64, device=values.device)
        splits = torch.zeros(row_lengths.numel() + 1, dtype=torch.int64, device=values.device)
        splits[1:] = torch.cumsum(row_lengths, 0)
        return cls(values, splits, validate=validate)

    @classmethod
    def from_numpy(cls, values, row_splits, device="cuda"):
        """Host arrays -> HBM (the contract of ``ragged_tensor_from_nested_numpy``, kgcnn/data/utils.py:129-157)."""
        v = torch.from_numpy(np.ascontiguousarray(values)).to(device)
        s = torch.from_numpy(np.ascontiguousarray(row_splits, dtype=np.int64)).to(device)
        out = cls(v, s)
        out._splits_host = np.asarray(row_splits, dtype=np.int64).copy()
        return out

    @classmethod
    def from_nested(cls, rows, dtype, inner_shape=(), device="cuda"):
        lens = [len(r) for r in rows]
        if sum(lens) == 0:
            vals = np.zeros((0,) + tuple(inner_shape), dtype=dtype)
        else:
            vals = np.concatenate([np.asarray(r, dtype=dtype).reshape((len(r),) + tuple(inner_shape))
                                   for r in rows], axis=0)
        splits = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        return cls.from_numpy(vals, splits, device=device)

    # ---- tf.RaggedTensor accessors --------------------------------------------------------------------------
    @property
    def flat_values(self):
        return self.values

    @property
    def dtype(self):
        return self.values.dtype

    @property
    def device(self):
        return self.values.device

    @property
    def shape(self):
        return (self.nrows(), None) + tuple(self.values.shape[1:])

    def nrows(self):
        return int(self.row_splits.shape[0]) - 1

    def row_lengths(self):
        return self.row_splits[1:] - self.row_splits[:-1]

    def row_splits_host(self):
        """Host copy of row_splits (one D2H copy, cached)."""
        if self._splits_host is None:
            self._splits_host = self.row_splits.cpu().numpy()
        return self._splits_host

    def value_rowids(self):
        lens = self.row_lengths()
        return torch.repeat_interleave(torch.arange(lens.numel(), device=lens.device), lens)

    def with_values(self, values):
        out = RaggedTensor(values, self.row_splits)
        out._splits_host = self._splits_host
        return out

    with_flat_values = with_values

    def __getitem__(self, i):
        s = self.row_splits_host()
        return self.values[int(s[i]):int(s[i + 1])]

    def numpy_rows(self):
        s = self.row_splits_host()
        v = self.values.cpu().numpy()
        return [v[s[i]:s[i + 1]] for i in range(len(s) - 1)]

    def to_tensor(self):
        from .layers.casting import ChangeTensorType
        return ChangeTensorType(input_tensor_type="ragged", output_tensor_type="tensor")(self)

    def __repr__(self):
        return "<RaggedTensor shape=%s dtype=%s device=%s>" % (self.shape, self.dtype, self.device)

    # ---- index plan -------------------------------------------------------------------------------------------
    def attach_plan(self, nodes, plan):
        self._plans[(nodes.row_splits.data_ptr(), int(nodes.values.shape[0]))] = plan

    def index_plan(self, nodes):
        """Plan of this (batch, [M], K) index tensor against the node partition of ``nodes``."""
        key = (nodes.row_splits.data_ptr(), int(nodes.values.shape[0]))
        plan = self._plans.get(key)
        if plan is None:
            plan = IndexPlan(self, nodes)
            self._plans[key] = plan
        return plan


class IndexPlan:
    """Shifted int32 index columns + lazily built CSR / stable-sort permutation per column.

    Replaces, per batch instead of per layer call: ``partition_row_indexing`` (kgcnn/ops/partition.py:140-155),
    the column pick ``shiftind[:, pooling_index]`` (kgcnn/layers/pooling.py:63), ``tf.argsort(stable=True)``
    (pooling.py:66) and the implicit segment boundaries of ``tf.math.segment_*``.
    """

    def __init__(self, idx, nodes):
        _ffi.require_device(idx.values, nodes.row_splits)
        if idx.values.dtype != torch.int64:
            raise TypeError("edge indices must be int64 (kgcnn/literature/Schnet.py:28), got %s" % idx.values.dtype)
        if idx.values.dim() != 2:
            raise ValueError("edge indices must have shape (batch, [M], K)")
        if idx.nrows() != nodes.nrows():
            raise ValueError("index tensor and node tensor have different batch sizes")
        self.M = int(idx.values.shape[0])
        self.K = int(idx.values.shape[1])
        self.N = int(nodes.values.shape[0])
        self.G = idx.nrows()
        dev = idx.values.device
        self.cols = torch.empty((self.K, max(self.M, 1)), dtype=torch.int32, device=dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        vals = idx.values.contiguous()
        _ffi.call("mp_index_prepare_i64", _ffi.ptr(vals), self.M, self.K, _ffi.ptr(nodes.row_splits),
                  _ffi.ptr(idx.row_splits), self.G, self.N, _ffi.ptr(self.cols), _ffi.ptr(self.flags), _ffi.stream())
        self._flags_host = None
        self._csr = {}

    @classmethod
    def from_prepared(cls, idx, nodes, cols, csr_ptr=None):
        """Plan for indices whose shifted int32 columns (and receiver CSR) a producer kernel already wrote
        (on-GPU ``SetRange``): range-checked and receiver-sorted by construction, nothing to launch."""
        self = cls.__new__(cls)
        self.M, self.K = int(idx.values.shape[0]), int(idx.values.shape[1])
        self.N, self.G = int(nodes.values.shape[0]), idx.nrows()
        self.cols = cols
        self.flags = torch.zeros(1, dtype=torch.int32, device=cols.device)
        self._flags_host = _ffi.MP_FLAG_UNSORTED_COL1  # column 0 sorted; column 1 not claimed
        self._csr = {}
        if csr_ptr is not None:
            self._csr[(0, False)] = self._csr[(0, True)] = (csr_ptr, None, cols[0, :self.M])
        return self

    @classmethod
    def from_host(cls, idx, nodes, plan, stream=None):
        """Plan computed by the host packer (``mp_pack_edge_index_host``) while it concatenated the batch: the shifted
        columns, flag word and CSR arrive with the batch instead of being recomputed by ``mp_index_prepare_i64``."""
        import ctypes
        self = cls.__new__(cls)
        self.M, self.K, self.N, self.G = plan["M"], plan["K"], plan["N"], idx.nrows()
        dev = idx.values.device
        st = _ffi.stream() if stream is None else stream
        self.cols = torch.empty((self.K, max(self.M, 1)), dtype=torch.int32, device=dev)
        if self.M:
            _ffi.call("mp_memcpy_h2d_async", _ffi.ptr(self.cols), plan["cols"].ctypes.data_as(ctypes.c_void_p),
                      self.K * self.M * 4, st)
        self.flags = torch.full((1,), plan["flags"], dtype=torch.int32, device=dev)
        self._flags_host = plan["flags"]
        self._csr = {}
        if plan.get("csr") is not None and not (plan["flags"] & _ffi.MP_FLAG_UNSORTED_COL0):
            ptr = torch.empty(self.N + 1, dtype=torch.int32, device=dev)
            _ffi.call("mp_memcpy_h2d_async", _ffi.ptr(ptr), plan["csr"].ctypes.data_as(ctypes.c_void_p),
                      (self.N + 1) * 4, st)
            self._csr[(0, False)] = self._csr[(0, True)] = (ptr, None, self.cols[0, :self.M])
        return self

    def flags_host(self):
        if self._flags_host is None:
            self._flags_host = int(self.flags.item())
        return self._flags_host

    def validate(self):
        """Opt-in range check (the analogue of TF-CPU's InvalidArgumentError on an out-of-range gather)."""
        if self.flags_host() & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")

    def col(self, k):
        if not 0 <= k < self.K:
            raise ValueError("index column %d out of range for K=%d" % (k, self.K))
        return self.cols[k, :self.M]

    def is_sorted(self, k):
        if k > 1:
            return False
        return not (self.flags_host() & (_ffi.MP_FLAG_UNSORTED_COL0 if k == 0 else _ffi.MP_FLAG_UNSORTED_COL1))

    def csr(self, k, assume_sorted=False):
        """(ptr, perm): CSR offsets over column k; perm is None when the column is already sorted."""
        key = (k, bool(assume_sorted))
        hit = self._csr.get(key)
        if hit is not None:
            return hit
        dev = self.cols.device
        ptr = torch.empty(self.N + 1, dtype=torch.int32, device=dev)
        seg = self.col(k)
        perm = None
        if not assume_sorted and not self.is_sorted(k) and self.M > 0:
            import ctypes
            nbytes = ctypes.c_size_t(0)
            _ffi.call("mp_sort_workspace_bytes", self.M, ctypes.byref(nbytes))
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            seg_sorted = torch.empty(self.M, dtype=torch.int32, device=dev)
            perm = torch.empty(self.M, dtype=torch.int32, device=dev)
            _ffi.call("mp_sort_segments_i32", _ffi.ptr(seg.contiguous()), self.M, _ffi.ptr(seg_sorted), _ffi.ptr(perm),
                      _ffi.ptr(ws), nbytes.value, _ffi.stream())
            seg = seg_sorted
        _ffi.call("mp_csr_from_sorted_i32", _ffi.ptr(seg.contiguous()) if self.M > 0 else None, self.M, self.N,
                  _ffi.ptr(ptr), _ffi.stream())
        out = (ptr, perm, seg)
        self._csr[key] = out
        return out
