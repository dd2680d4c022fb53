"""Result buffers of a batch slot.

A Keras model call returns a NEW tensor.  A captured HIP graph writes to fixed addresses, so the fused routes used to copy
the slot's static result into a fresh allocation after every replay - a launch of its own per result (4.3 us of a 64 us
SchNet forward at BASELINE config 2).  Instead a slot keeps a few result sets, each with its own captured graph (the same
kernels on the same work buffers - only the destination of the final kernels differs), and hands a set out only while
nobody else holds one of its tensors, a view of them or their storage: storage use count and the Python reference counts
of the tensor and of its storage wrapper are back at the values they had when only the ring held the tensor.  To the caller that is indistinguishable from a fresh tensor - a
result somebody still holds is never written again - and a loop that drops its results runs on the ring alone.  When
every set is held the caller falls back to a static set and a copy.
"""
import sys

import torch

from . import _ffi


_USE_COUNT = getattr(torch._C, "_storage_Use_Count", None)   # private torch API (also used by torch's CUDA-graph trees)


def _holders(t):
    """(storage use count, references to the tensor object, references to its Python storage wrapper).  The third catches
    a caller who keeps only ``out.untyped_storage()``: torch preserves and re-uses the wrapper object, so the first two
    counts stay at their idle values while the wrapper's own reference count is one higher."""
    st = t.untyped_storage()
    return _USE_COUNT(st._cdata), sys.getrefcount(t), sys.getrefcount(st)


class ResultRing:

    def __init__(self, size=3):
        self.size = int(size)
        self._entries, self._next = [], 0      # entry: [tensors, graph, idle holder counts]

    def acquire(self, make, capture):
        """``(tensors, graph)`` of a result set no caller holds, or ``None`` when all are held.  ``make() -> tuple of
        tensors`` allocates a set, ``capture(tensors) -> graph`` captures the slot's launches writing into it (both
        passed per call, so the ring holds no reference back to its slot)."""
        if _USE_COUNT is None:      # no way to tell whether a view of a buffer is still alive: always copy
            return None
        ring, entry = self._entries, None
        # The ring is filled to its size before a set is used a second time, and idle sets are then taken round robin:
        # consecutive calls on one slot launch DIFFERENT graph executables.  Launching an executable whose previous launch
        # is still running makes hipGraphLaunch wait for it on the host (its kernel arguments are still in use) - with one
        # result set per slot the host could never be more than one forward ahead per stream, and while it waited for one
        # stream's forward it fed none of the others.
        if len(ring) >= self.size:
            for k in range(len(ring)):
                cand = ring[(self._next + k) % len(ring)]
                if all(_holders(cand[0][i]) == cand[2][i] for i in range(len(cand[0]))):
                    entry = cand
                    self._next = (self._next + k + 1) % len(ring)
                    break
            if entry is None:
                return None
        else:
            entry = [tuple(make()), None, None]
            ring.append(entry)
            entry[2] = [_holders(entry[0][i]) for i in range(len(entry[0]))]
            self._next = 0
        if entry[1] is None:
            entry[1] = capture(entry[0])
        return entry[0], entry[1]

    def destroy(self):
        for entry in self._entries:
            if entry[1] is not None:
                try:
                    _ffi.call("mp_graph_destroy", entry[1])
                except Exception:
                    pass
        self._entries = []
