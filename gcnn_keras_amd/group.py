"""Union of several resident ragged batches on the device (launch groups).

``BatchUnion([inputs_1, ..., inputs_k])`` owns the concatenated tensors of k independent ``[node_number, coordinates,
edge_indices]`` input sets and the descriptor of ``mp_concat_batches`` (csrc/mp_concat.hip: ONE launch copies node numbers,
coordinates and edge sample indices and rebases the row splits - the disjoint union the reference forms when it concatenates
graph lists, kgcnn/data/base.py:203-239; sample indices need no rewriting, kgcnn/layers/base.py:27).  A fused route binds an
ordinary batch slot on ``union.inputs`` and runs ``union.concat`` in front of every forward (inside the captured graph), so
k batches are served by one launch sequence and new values in a member's tensors are picked up.  Graphs of a disjoint batch
do not interact: every member's rows equal a forward of its own up to the rounding order of boundary sums."""
import ctypes

import numpy as np
import torch

from . import _ffi
from .ragged import RaggedTensor


class BatchUnion:

    def __init__(self, inputs_list):
        k = len(inputs_list)
        if not 1 <= k <= _ffi.MP_CONCAT_MAX:
            raise ValueError("a launch group holds 1..%d batches" % _ffi.MP_CONCAT_MAX)
        self.members = [tuple(x) for x in inputs_list]          # keeps the member tensors (and their addresses) alive
        z0 = inputs_list[0][0].values
        if any(x[0].values.dtype != z0.dtype for x in inputs_list):
            raise ValueError("the members of a launch group must share the node-number dtype")
        ns_host = [np.asarray(x[0].row_splits_host(), dtype=np.int64) for x in inputs_list]
        es_host = [np.asarray(x[2].row_splits_host(), dtype=np.int64) for x in inputs_list]
        self.sizes = [(int(x[0].values.shape[0]), int(x[2].values.shape[0]), x[0].nrows()) for x in inputs_list]
        n, m, g = (sum(t[i] for t in self.sizes) for i in range(3))
        self.N, self.M, self.G = n, m, g
        dev = z0.device
        z = torch.empty(n, dtype=z0.dtype, device=dev)
        xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
        idx = torch.empty((m, 2), dtype=torch.int64, device=dev)
        ns = torch.empty(g + 1, dtype=torch.int64, device=dev)
        es = torch.empty(g + 1, dtype=torch.int64, device=dev)
        d = _ffi.ConcatDesc()
        d.k, d.z_is_i64 = k, 1 if z0.dtype == torch.int64 else 0
        for b, (x, (nb, mb, gb)) in enumerate(zip(inputs_list, self.sizes)):
            node, xyz_b, idx_b = x
            src = d.src[b]
            src.z, src.xyz, src.idx = node.values.data_ptr(), xyz_b.values.data_ptr(), idx_b.values.data_ptr()
            src.node_splits, src.edge_splits = node.row_splits.data_ptr(), idx_b.row_splits.data_ptr()
            src.N, src.M, src.G = nb, mb, gb
        d.z, d.xyz, d.idx = z.data_ptr(), xyz.data_ptr(), idx.data_ptr()
        d.node_splits, d.edge_splits = ns.data_ptr(), es.data_ptr()
        self.desc, self._desc_ref = d, ctypes.byref(d)
        # host splits of the union; per member its graph range cut at ITS OWN row count (graphs without nodes at a member's
        # end are dropped from its result, as its own forward would drop them) and its node range
        n_off = m_off = g_off = 0
        cat_n, cat_e = [np.zeros(1, np.int64)], [np.zeros(1, np.int64)]
        self.graph_cuts, self.node_cuts = [], []
        for hn, he, (nb, mb, gb) in zip(ns_host, es_host, self.sizes):
            cat_n.append(hn[1:] + n_off)
            cat_e.append(he[1:] + m_off)
            rows = gb
            while rows > 0 and hn[rows] == hn[rows - 1]:
                rows -= 1
            self.graph_cuts.append((g_off, g_off + rows))
            self.node_cuts.append((n_off, n_off + nb))
            n_off, m_off, g_off = n_off + nb, m_off + mb, g_off + gb

        def rag(values, splits, host):
            r = RaggedTensor(values, splits)
            r._splits_host = host
            return r

        hn, he = np.concatenate(cat_n), np.concatenate(cat_e)
        node_r = rag(z, ns, hn)
        xyz_r = rag(xyz, ns, hn)
        self.inputs = [node_r, xyz_r, rag(idx, es, he)]

    def concat(self):
        _ffi.check(_ffi.lib().mp_concat_batches(self._desc_ref, _ffi.stream()))

    def split_graphs(self, out):
        """Per-member views of a per-graph result ``(rows, L)`` of the union."""
        full = int(out.shape[0])
        return [out[lo:min(hi, full)] for lo, hi in self.graph_cuts]

    def split_nodes(self, out):
        """Per-member views of a per-node result ``(N, ...)`` of the union."""
        return [out[lo:hi] for lo, hi in self.node_cuts]
