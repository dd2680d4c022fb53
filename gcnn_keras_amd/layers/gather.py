"""Gather layers (mirror of kgcnn/layers/gather.py) on the HIP engine.

Same class names, constructor arguments, ``call([nodes, indices])`` conventions and ``get_config`` keys as the
reference; the TF op sequence ``partition_row_indexing -> tf.gather -> slices -> tf.concat`` is one kernel
(``mp_gather_rows_f32``) over index columns that were shifted once per batch.
"""
import torch

from .. import _ffi
from ..ragged import RaggedTensor
from .base import GraphBaseLayer


def gather_rows(values, plan, colsel):
    """rows of ``values`` (N, ...) at plan columns ``colsel`` -> (M, len(colsel), ...) contiguous."""
    from ..autograd import GatherRows, needs_grad
    if needs_grad(values):
        return GatherRows.apply(values, plan, tuple(colsel))
    return _gather_rows_raw(values, plan, colsel)


def _gather_rows_raw(values, plan, colsel):
    _ffi.require_device(values)
    vals = values.contiguous()
    elems = 1
    for d in vals.shape[1:]:
        elems *= int(d)
    elems = max(elems, 1)
    out = torch.empty((plan.M, len(colsel)) + tuple(vals.shape[1:]), dtype=vals.dtype, device=vals.device)
    if vals.dtype != torch.float32:
        raise TypeError("gather expects float32 node values, got %s" % vals.dtype)
    _ffi.call("mp_gather_rows_f32", _ffi.ptr(vals), int(vals.shape[0]), elems, _ffi.ptr(plan.cols), plan.M,
              len(colsel), _ffi.int32_array(list(colsel)), _ffi.ptr(out), _ffi.stream())
    return out


class GatherEmbedding(GraphBaseLayer):
    r"""Gather node embeddings for every index of an edge ``(i, j)``; default output ``[x_i || x_j]`` of shape
    ``(batch, [M], 2*F)`` (kgcnn/layers/gather.py:9-149).  The reference's disjoint fast path (gather.py:69-99:
    ``axis == 1``, concat / split axis ``None`` or 2) is one gather kernel; other concat / split axes take the general
    route of gather.py:121-138 (same gather + strided re-arrangement); only ``axis != 1`` (ragged rank 2 in TF) raises."""

    def __init__(self, axis: int = 1, concat_axis: int = 2, split_axis: int = None, split_indices: list = None,
                 concat_indices: list = None, node_indexing: str = "sample", **kwargs):
        super().__init__(node_indexing=node_indexing, **kwargs)
        self.concat_axis = concat_axis
        self.axis = axis
        self.split_axis = split_axis
        self.split_indices = split_indices
        self.concat_indices = concat_indices
        if split_axis is not None and concat_axis is not None:
            raise ValueError("Can not both split and concatenate new index axis. At least one must be `None`.")

    def build(self, input_shape):
        super().build(input_shape)
        if len(input_shape) != 2:
            print("Number of inputs for layer '%s' must be 2: `[nodes, indices]` ." % self.name)

    def call(self, inputs, **kwargs):
        r"""inputs: ``[embeddings (batch, [N], F), tensor_index (batch, [M], K)]``."""
        nodes, idx = self.assert_ragged_input_rank(list(inputs))
        if self.axis != 1:
            # tf.gather(batch_dims=1, axis > 1) picks FEATURE positions per edge and yields ragged rank 2 - no kgcnn model
            # uses it, and this package's ragged carrier is rank 1
            raise NotImplementedError("GatherEmbedding gathers along the node axis (axis=1)")
        plan = idx.index_plan(nodes)
        if self.ragged_validate:
            plan.validate()
        if self.concat_axis not in [None, 2] or self.split_axis not in [None, 2]:
            return self._general(nodes, idx, plan)
        if self.concat_axis == 2:
            cols = list(self.concat_indices) if self.concat_indices else list(range(plan.K))
            out = gather_rows(nodes.values, plan, cols)  # (M, K, F...)
            # tf.concat([out[:, i] ...], axis=1): (M, K, F) is already [x_i || x_j] row-major
            out = out.reshape((plan.M, len(cols) * int(out.shape[2])) + tuple(out.shape[3:])) if out.dim() > 2 \
                else out.reshape(plan.M, len(cols))
            return idx.with_values(out)
        if self.split_axis == 2:
            cols = list(self.split_indices) if self.split_indices else list(range(plan.K))
            return [idx.with_values(gather_rows(nodes.values, plan, [c])[:, 0].contiguous()) for c in cols]
        return idx.with_values(gather_rows(nodes.values, plan, list(range(plan.K))))

    def _general(self, nodes, idx, plan):
        """The reference's general route (gather.py:121-138): ``out = tf.gather(nodes, index, batch_dims=1, axis=1)`` of
        shape ``(batch, [M], K, F...)``, then ``tf.gather(out, i, axis)`` for every concat / split index along ANY later
        axis and ``tf.concat`` along that same axis.  The gather is the engine kernel; picking index ``i`` along a later
        axis and re-joining are strided copies of the gathered values (layout only)."""
        out = gather_rows(nodes.values, plan, list(range(plan.K)))        # values of (batch, [M], K, F...)
        rank = out.dim() + 1                                              # rank of the ragged tensor incl. batch axis

        def positive(axis):
            axis = axis + rank if axis < 0 else axis
            if not 2 <= axis < rank:
                raise ValueError("axis %d is not a dense axis of the gathered tensor of rank %d" % (axis, rank))
            return axis - 1                                               # same axis on the values tensor

        if self.concat_axis is not None:
            ax = positive(self.concat_axis)
            picks = list(self.concat_indices) if self.concat_indices else list(range(int(out.shape[ax])))
            parts = [out.select(ax, i) for i in picks]                    # tf.gather(out, i, axis): drops the axis
            if ax >= parts[0].dim():   # tf.concat raises the same way: the picked tensors lost that axis
                raise ValueError("concat_axis %d is past the last axis once an index is picked along it" % self.concat_axis)
            return idx.with_values(torch.cat(parts, dim=ax).contiguous())
        ax = positive(self.split_axis)
        picks = list(self.split_indices) if self.split_indices else list(range(int(out.shape[ax])))
        return [idx.with_values(out.select(ax, i).contiguous()) for i in picks]

    def get_config(self):
        config = super().get_config()
        config.update({"concat_axis": self.concat_axis, "axis": self.axis, "split_axis": self.split_axis,
                       "concat_indices": self.concat_indices, "split_indices": self.split_indices,
                       "node_indexing": self.node_indexing})
        return config


GatherNodes = GatherEmbedding


class GatherEmbeddingSelection(GraphBaseLayer):
    r"""Gather embeddings for the given index columns; always returns a list (kgcnn/layers/gather.py:153-245)."""

    def __init__(self, selection_index, axis: int = 1, axis_indices: int = 2, **kwargs):
        super().__init__(**kwargs)
        self.axis = axis
        self.axis_indices = axis_indices
        if not isinstance(selection_index, (list, tuple, int)):
            raise ValueError("Indices for selection must be list or tuple for layer `GatherEmbeddingSelection`.")
        self.selection_index = [selection_index] if isinstance(selection_index, int) else list(selection_index)

    def call(self, inputs, **kwargs):
        nodes, idx = self.assert_ragged_input_rank(list(inputs))
        if self.axis != 1 or self.axis_indices != 2:
            raise NotImplementedError("Only the disjoint fast path (axis=1, axis_indices=2) is built.")
        plan = idx.index_plan(nodes)
        if self.ragged_validate:
            plan.validate()
        return [idx.with_values(gather_rows(nodes.values, plan, [i])[:, 0]) for i in self.selection_index]

    def get_config(self):
        config = super().get_config()
        config.update({"axis": self.axis, "axis_indices": self.axis_indices, "selection_index": self.selection_index})
        return config


GatherNodesSelection = GatherEmbeddingSelection


class GatherNodesIngoing(GatherEmbeddingSelection):
    r"""Gather the receiving node ``i`` of each edge ``(i, j)`` (kgcnn/layers/gather.py:249-282)."""

    def __init__(self, selection_index: int = 0, **kwargs):
        super().__init__(selection_index=selection_index, **kwargs)

    def call(self, inputs, **kwargs):
        return super().call(inputs, **kwargs)[0]


class GatherNodesOutgoing(GatherEmbeddingSelection):
    r"""Gather the sending node ``j`` of each edge ``(i, j)`` (kgcnn/layers/gather.py:286-319)."""

    def __init__(self, selection_index: int = 1, **kwargs):
        super().__init__(selection_index=selection_index, **kwargs)

    def call(self, inputs, **kwargs):
        return super().call(inputs, **kwargs)[0]


class GatherState(GraphBaseLayer):
    r"""Repeat a per-graph state for every node / edge of its graph (kgcnn/layers/gather.py:323-375)."""

    def call(self, inputs, **kwargs):
        env, target = inputs[0], inputs[1]
        if not isinstance(target, RaggedTensor):
            target = self.assert_ragged_input_rank(target)
        _ffi.require_device(env, target.row_splits)
        envc = env.contiguous()
        elems = 1
        for d in envc.shape[1:]:
            elems *= int(d)
        n = int(target.values.shape[0])
        out = torch.empty((n,) + tuple(envc.shape[1:]), dtype=envc.dtype, device=envc.device)
        _ffi.call("mp_repeat_rows_f32", _ffi.ptr(envc), _ffi.ptr(target.row_splits), target.nrows(), max(elems, 1), n,
                  _ffi.ptr(out), _ffi.stream())
        return target.with_values(out)
