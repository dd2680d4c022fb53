"""Host batch packer: lists of per-graph NumPy arrays -> the engine's ragged tensors in HBM.

Mirrors ``MemoryGraphList.tensor`` (kgcnn/data/base.py:203-239: per property ``np.concatenate`` + ``row_lengths``
through ``ragged_tensor_from_nested_numpy``, kgcnn/data/utils.py:129-157).  Rows are concatenated straight into reusable
staging buffers - pinned when a GPU is present - and every packed tensor then crosses PCIe with one asynchronous copy.
For the edge indices the native library (``mp_pack_edge_index_host``, csrc/mp_pack.hip) makes the batch's index plan in
the same pass (shifted int32 columns, flag word, CSR), so that the device does not have to recompute it
(``IndexPlan.from_host``).

Gathering many small per-graph arrays is bound by the per-array Python work, not by bytes: collecting 3 x 128 array
addresses for a native pointer table cost 1.1 ms per 128-graph batch (``a.ctypes.data`` ~1 us each, plus the per-graph
``ascontiguousarray``), three times the GPU time of 30 such batches.  ``np.concatenate(..., out=staging)`` walks the list in
C (~50 ns per array); the native passes then run on the concatenated block through a pointer table made by vectorised
arithmetic on the row splits (``mp_pack_rows_host`` stays the entry point for callers that hold a pointer table already).
"""
import ctypes

import numpy as np
import torch

from .. import _ffi
from ..ragged import IndexPlan, RaggedTensor

_KINDS = {np.dtype("float32"): 0, np.dtype("float64"): 1, np.dtype("int32"): 2, np.dtype("int64"): 3}


def _kind(dtype):
    dtype = np.dtype(dtype)
    if dtype not in _KINDS:
        raise TypeError("the packer handles float32/float64/int32/int64 properties, got %s" % dtype)
    return _KINDS[dtype]


class HostBuffer:
    """Growable staging block from ``mp_host_alloc`` (pinned if a device is present), exposed as NumPy views."""

    def __init__(self, pinned=None):
        self.pinned = _ffi.has_gpu() if pinned is None else bool(pinned)
        self._ptr = ctypes.c_void_p(None)
        self._bytes = 0

    def view(self, shape, dtype):
        dtype = np.dtype(dtype)
        need = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if need > self._bytes:
            self.release()
            # half again as much as asked for: batches of a dataset differ by ~10 % in size, and a pinned allocation costs
            # ~0.4 ms (more than packing the batch) - one allocation per staging block, not one per new maximum
            size = max(need + need // 2, 2 * self._bytes, 1 << 16)
            _ffi.call("mp_host_alloc", size, int(self.pinned), ctypes.byref(self._ptr))
            self._bytes = size
        if need == 0:
            return np.zeros(shape, dtype=dtype)
        raw = (ctypes.c_char * need).from_address(self._ptr.value)
        raw._owner = self  # views keep their staging block alive (np.frombuffer holds `raw`)
        return np.frombuffer(raw, dtype=dtype).reshape(shape)

    def release(self):
        """Frees the block; views handed out earlier must not be used afterwards (growth replaces the block only when a
        larger view is requested, i.e. when the caller starts the next batch in this slot)."""
        if self._ptr.value:
            _ffi.call("mp_host_free", self._ptr, int(self.pinned))
            self._ptr = ctypes.c_void_p(None)
            self._bytes = 0

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _row_table(arrays, src_dtype, inner_shape):
    keep = [np.ascontiguousarray(a, dtype=src_dtype).reshape((len(a),) + inner_shape) for a in arrays]
    ptrs = (ctypes.c_void_p * len(keep))(*[a.ctypes.data if a.size else None for a in keep])
    counts = np.array([a.shape[0] for a in keep], dtype=np.int64)
    return keep, ptrs, counts


def _concat_into(arrays, out):
    """``np.concatenate(arrays, axis=0)`` into ``out`` (any dtype conversion included); arrays without rows may have any
    shape (the reference's datasets hold ``(0,)`` placeholders)."""
    try:
        np.concatenate(arrays, axis=0, out=out, casting="unsafe")
    except ValueError:
        kept = [np.asarray(a).reshape((len(a),) + out.shape[1:]) for a in arrays if len(a)]
        if kept:
            np.concatenate(kept, axis=0, out=out, casting="unsafe")


def pack_rows(arrays, dtype=None, buffer=None, threads=4):
    """``(values, row_splits)`` on the host: ``np.concatenate(arrays, axis=0, dtype=dtype)`` + int64 row_splits."""
    arrays = list(arrays)
    G = len(arrays)
    first = next((np.asarray(a) for a in arrays if len(a)), np.asarray(arrays[0]) if G else np.zeros((0,)))
    inner = tuple(first.shape[1:])
    src = first.dtype if G else np.dtype("float32")
    dst = np.dtype(dtype) if dtype is not None else src
    if dtype is not None:
        _kind(dst)                                   # float32 / float64 / int32 / int64 targets (TypeError otherwise)
    elif dst not in _KINDS:                          # e.g. int16 / bool properties keep a widened type, as before
        dst = np.dtype("int64") if dst.kind in "iub" else np.dtype("float64")
    if src.kind == "f" and dst.kind in "iu":
        raise _ffi.EngineError("mp_pack_rows_host: float -> integer is not a conversion the packer offers")
    counts = np.fromiter((len(a) for a in arrays), dtype=np.int64, count=G)
    total = int(counts.sum())
    buffer = buffer or HostBuffer(pinned=False)
    values = buffer.view((total,) + inner, dst)
    splits = np.zeros(G + 1, dtype=np.int64)
    np.cumsum(counts, out=splits[1:])
    if total:
        for a in arrays[:1] + arrays[-1:]:
            a = np.asarray(a)
            if a.size and tuple(a.shape[1:]) != inner:
                raise ValueError("all arrays must match in shape except the first dimension (kgcnn/data/utils.py:130-131)")
        try:
            _concat_into(arrays, values)
        except ValueError:
            raise ValueError("all arrays must match in shape except the first dimension (kgcnn/data/utils.py:130-131)")
    return values, splits


def to_device(values, splits, device="cuda", stream=None):
    """One asynchronous copy per array; the ragged tensor keeps the host splits (no D2H later)."""
    dev_vals = torch.empty(values.shape, dtype=torch.from_numpy(np.zeros(1, values.dtype)).dtype, device=device)
    dev_splits = torch.empty(splits.shape, dtype=torch.int64, device=device)
    st = _ffi.stream() if stream is None else stream
    _ffi.call("mp_memcpy_h2d_async", _ffi.ptr(dev_vals), values.ctypes.data_as(ctypes.c_void_p), values.nbytes, st)
    _ffi.call("mp_memcpy_h2d_async", _ffi.ptr(dev_splits), splits.ctypes.data_as(ctypes.c_void_p), splits.nbytes, st)
    out = RaggedTensor(dev_vals, dev_splits)
    out._splits_host = splits.copy()
    out._staging_view = values  # valid until this staging slot is packed again (used for topology signatures)
    return out


def pack_edge_index(index_arrays, node_counts, buffers=None, threads=4, with_csr=True):
    """Host pass over the per-graph ``(m_g, K)`` index lists -> dict of host arrays: ``idx`` (M,K) int64 (the API tensor),
    ``edge_splits``, ``node_splits``, ``cols`` (K,M) int32 shifted, ``csr`` (N+1) int32, ``flags`` int."""
    index_arrays = list(index_arrays)
    G = len(index_arrays)
    first = next((np.asarray(a) for a in index_arrays if len(a)), None)
    K = int(first.shape[1]) if first is not None else 2
    if first is not None and first.dtype.kind not in "iu":
        raise TypeError("the packer handles float32/float64/int32/int64 properties, edge indices must be integers")
    ecounts = np.fromiter((len(a) for a in index_arrays), dtype=np.int64, count=G)
    ncounts = np.ascontiguousarray(node_counts, dtype=np.int64)
    if ncounts.shape != (G,):
        raise ValueError("node_counts must have one entry per graph")
    M, N = int(ecounts.sum()), int(ncounts.sum())
    buffers = buffers or {}
    def hb(name):
        if name not in buffers:
            buffers[name] = HostBuffer(pinned=False)
        return buffers[name]

    idx = hb("idx").view((M, K), np.int64)
    cols = hb("cols").view((K, max(M, 1)), np.int32)
    csr = hb("csr").view((N + 1,), np.int32) if with_csr else None
    esplits = np.zeros(G + 1, dtype=np.int64)
    nsplits = np.zeros(G + 1, dtype=np.int64)
    flags = ctypes.c_int32(0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None
    table = None
    if M:
        # the API tensor first (C loop over the list), then the native plan pass reads it IN PLACE through a pointer table
        # made from the row splits: graph g's rows start at idx + 8 K * edge_splits[g]
        _concat_into(index_arrays, idx)
        table = (np.cumsum(ecounts) - ecounts).astype(np.uint64) * np.uint64(8 * K) + np.uint64(idx.ctypes.data)
    _ffi.call("mp_pack_edge_index_host", vp(table), _kind(np.int64), vp(ecounts), vp(ncounts), G, K, vp(idx), vp(esplits),
              vp(nsplits), vp(cols) if M else None, vp(csr), ctypes.byref(flags), int(threads))
    return {"idx": idx, "edge_splits": esplits, "node_splits": nsplits, "cols": cols, "csr": csr,
            "flags": int(flags.value), "M": M, "N": N, "K": K}


class PackedBatch(dict):
    """Name -> device tensor (``RaggedTensor`` for ragged items, ``torch.Tensor`` otherwise) plus ``.ready``, a
    ``torch.cuda.Event`` recorded on the copy stream after the last copy of the batch.

    Every device tensor of the batch is allocated on the packer's copy stream, so the caching allocator would hand its
    memory to the NEXT ``pack()`` as soon as the batch is dropped - while kernels of the consumer stream may still be
    queued on it.  ``wait(stream)`` therefore does both halves of the hand-over: the consumer stream waits for the copies,
    and every tensor (values, row_splits, the plan's columns / CSR / flag word) is registered with the consumer stream
    (``Tensor.record_stream``), which keeps its block out of the copy stream's pool until that stream has passed the
    point of release."""
    ready = None

    def device_tensors(self):
        seen = set()

        def emit(t):
            if torch.is_tensor(t) and t.is_cuda and id(t) not in seen:
                seen.add(id(t))
                yield t

        for value in self.values():
            if isinstance(value, RaggedTensor):
                yield from emit(value.values)
                yield from emit(value.row_splits)
                for plan in value._plans.values():
                    yield from emit(plan.cols)
                    yield from emit(plan.flags)
                    for entry in plan._csr.values():
                        for t in entry:
                            yield from emit(t)
            else:
                yield from emit(value)

    def wait(self, stream=None):
        stream = stream or torch.cuda.current_stream()
        if self.ready is not None:
            stream.wait_event(self.ready)
        for t in self.device_tensors():
            t.record_stream(stream)
        return self


class BatchPacker:
    """``MemoryGraphList.tensor(items)`` (kgcnn/data/base.py:219-239) for a list of graph dicts, double-buffered:
    batch ``k+1`` is packed and copied on a side stream while batch ``k`` computes.

    ``items`` are the reference's input descriptors, e.g. ``{"name": "edge_indices", "ragged": True, "dtype": "int64"}``.
    ``index_item`` names the edge-index property whose plan is built on the host; ``node_item`` the property that
    defines the node partition (its ragged tensor receives the plan via ``attach_plan``)."""

    def __init__(self, items, index_item="edge_indices", node_item=None, device="cuda", threads=4, slots=2):
        self.items = [dict(it) for it in (items.values() if isinstance(items, dict) else items)]
        self.keys = list(items.keys()) if isinstance(items, dict) else [it["name"] for it in self.items]
        names = [it["name"] for it in self.items]
        self.index_item = index_item if index_item in names else None
        self.node_item = node_item or next((n for n in names if n != index_item and self._is_ragged(n)), None)
        self.device = device
        self.threads = threads
        self._slots = [{} for _ in range(slots)]
        self._turn = 0
        self._copy_stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self._slot_events = [None] * slots

    def _is_ragged(self, name):
        return bool(next(it for it in self.items if it["name"] == name).get("ragged", False))

    def pack_host(self, graphs, slot=None):
        """Host half only (no device needed): name -> ``(values, splits)`` or dense array; plan dict under ``"__plan__"``."""
        bufs = self._slots[self._turn if slot is None else slot]

        def hb(name):
            if name not in bufs:
                bufs[name] = HostBuffer()
            return bufs[name]

        out = {}
        for key, it in zip(self.keys, self.items):
            name = it["name"]
            props = [g[name] for g in graphs]
            dtype = it.get("dtype")
            if not it.get("ragged", False):
                out[key] = np.array(props, dtype=dtype)   # tf.constant(np.array(props)), kgcnn/data/base.py:216
            elif name == self.index_item:
                node_counts = [len(g[self.node_item]) for g in graphs]
                sub = {k: hb("%s/%s" % (name, k)) for k in ("idx", "cols", "csr")}
                plan = pack_edge_index(props, node_counts, buffers=sub, threads=self.threads)
                out[key] = (plan["idx"], plan["edge_splits"])
                out["__plan__"] = plan
            else:
                out[key] = pack_rows(props, dtype=dtype, buffer=hb(name), threads=self.threads)
        return out

    def pack(self, graphs):
        """Pack and ship one batch; returns a ``PackedBatch`` whose copies run on the packer's copy stream."""
        slot = self._turn
        if self._slot_events[slot] is not None:
            self._slot_events[slot].synchronize()   # the staging block of this slot is free again
        host = self.pack_host(graphs, slot)
        self._turn = (self._turn + 1) % len(self._slots)
        batch = PackedBatch()
        plan = host.pop("__plan__", None)
        with torch.cuda.stream(self._copy_stream):
            st = _ffi.stream()
            for key, it in zip(self.keys, self.items):
                if not it.get("ragged", False):
                    arr = np.ascontiguousarray(host[key])
                    t = torch.empty(arr.shape, dtype=torch.from_numpy(np.zeros(1, arr.dtype)).dtype, device=self.device)
                    _ffi.call("mp_memcpy_h2d_async", _ffi.ptr(t), arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes, st)
                    batch[key] = t
                    batch.setdefault("__keep__", []).append(arr)
                else:
                    batch[key] = to_device(*host[key], device=self.device, stream=st)
            if plan is not None and self.node_item is not None:
                idx_key = self.keys[[it["name"] for it in self.items].index(self.index_item)]
                node_key = self.keys[[it["name"] for it in self.items].index(self.node_item)]
                batch[idx_key].attach_plan(batch[node_key], IndexPlan.from_host(batch[idx_key], batch[node_key], plan, st))
            ev = torch.cuda.Event()
            ev.record()
        batch.ready = ev
        self._slot_events[slot] = ev
        return batch
