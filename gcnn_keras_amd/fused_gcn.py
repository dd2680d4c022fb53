"""Fused GCN forward: the arithmetic of kgcnn/literature/GCN.py:95-109 in 1 + depth launches, replayed from a HIP graph.

    input launch        n = Dense(units, linear)(node_attributes)  and  h = gcn[0].lay_dense(n)                (1 launch)
    per GCN layer i     n = act(sum_e w_e h[send(e)])  (gather, weighted pool, activation: gcn_conv.py:87-90)
                        and  h = gcn[i+1].lay_dense(n), or - after the last layer - out = GraphMLP(n)     (depth launches)

(csrc/mp_gcn.hip, ``mp_gcn_tile_f32``).  The route mirrors ``fused.SchnetFusedRoute``: it reads the model's own weight
tensors in place, keeps one batch slot per distinct input set (work buffers, index plan, captured graph), launches on
torch's current stream and hands every call a result nobody else holds.
"""
import ctypes

import torch

from . import _ffi
from .result_ring import ResultRing

_SUM = ("sum", "segment_sum", "reduce_sum")
_WIDTHS = (32, 64, 128)


def _as_list(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def supports(config):
    """True if a ``GCN.make_model`` configuration maps onto the tile kernel: node features given as vectors (no
    embedding), scalar edge weights, ``units`` in {32, 64, 128}, sum pooling without weight normalisation, an engine
    activation in the layers, node output through a GraphMLP of at most three Dense layers of width <= 128 whose last
    activation may be softmax.  Everything else runs the layer sequence."""
    try:
        inputs, ga, om = config["inputs"], config["gcn_args"], config["output_mlp"]
        units = _as_list(om["units"], 1)
        acts = _as_list(om.get("activation"), len(units))
        if len(inputs[0]["shape"]) != 2 or len(inputs[1]["shape"]) != 2 or int(inputs[1]["shape"][-1]) != 1:
            return False
        if any(om.get(k) and any(_as_list(om.get(k), len(units))) for k in ("use_dropout", "use_normalization")):
            return False
        return bool(
            config.get("output_embedding") == "node" and int(config["depth"]) >= 1
            and int(ga["units"]) in _WIDTHS and ga.get("pooling_method", "sum") in _SUM
            and not ga.get("normalize_by_weights", False) and ga.get("use_bias", True) in (True, False)
            and ga.get("activation", "kgcnn>leaky_relu") in _ffi.ACTIVATION_CODES
            and 1 <= len(units) <= 3 and all(1 <= int(u) <= 128 for u in units)
            and all(a in _ffi.ACTIVATION_CODES for a in acts[:-1])
            and (acts[-1] == "softmax" or acts[-1] in _ffi.ACTIVATION_CODES))
    except (KeyError, TypeError, IndexError, ValueError):
        return False


class FusedGcn:
    """One batch slot: the index plan, the two (N, units) buffers the layers alternate between, result buffers and the
    captured graphs of ONE bound input set."""

    def __init__(self, route, node, edge_w, idx):
        self.route = route
        self.node, self.edge_w, self.idx = node, edge_w, idx
        x = node.values
        self.N, self.K = int(x.shape[0]), int(x.shape[1])
        plan = idx.index_plan(node)
        lay = route.gcns[0].lay_pool
        ptr, perm, _ = plan.csr(lay.pooling_index, assume_sorted=lay.is_sorted)
        self.M = plan.M
        if plan.flags_host() & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")
        self.ptr, self.perm = ptr, perm
        self.tile_start, self.n_tiles = self._balanced_tiles(ptr)
        self.send = plan.col(1 - lay.pooling_index).contiguous()
        self.weight = edge_w.values.contiguous().view(-1)
        units = route.units
        self.h = [torch.empty((self.N, units), dtype=torch.float32, device=x.device) for _ in range(2)]
        self.out_units = route.out_units
        self._ring = ResultRing()
        self._static = None      # [out, graph] used when every ring buffer is held
        self.stream = torch.cuda.Stream()
        self.calls = 0

    TILE_EDGES = 256     # edges per aggregate tile the table aims at: one round of 16 rows for each of 16 thread groups

    def _balanced_tiles(self, ptr):
        """Tile table of the aggregate launches: at most 16 consecutive nodes and - unless a single node has more - at
        most ``TILE_EDGES`` edges per tile.  With 16 nodes per tile throughout, a hub-heavy stretch of nodes (the first
        16 nodes of a Barabasi-Albert graph hold ~1000 of Cora's 13264 edges) sets the launch's time alone.  ``None`` when
        the uniform tiles are balanced already (every molecular batch) or the graph is too large for a host pass."""
        import numpy as np
        n = self.N
        if n == 0 or n > 200000:
            return None, 0
        ptr_h = ptr.cpu().numpy().astype(np.int64)
        uniform = np.diff(ptr_h[np.minimum(np.arange(0, n + 16, 16), n)])
        if uniform.size == 0 or int(uniform.max()) <= 2 * self.TILE_EDGES:
            return None, 0
        deg = np.diff(ptr_h)
        starts, nodes, edges = [0], 0, 0
        for i in range(n):
            if nodes == 16 or (nodes > 0 and edges + int(deg[i]) > self.TILE_EDGES):
                starts.append(i)
                nodes, edges = 0, 0
            nodes += 1
            edges += int(deg[i])
        starts.append(n)
        table = torch.tensor(starts, dtype=torch.int32, device=ptr.device)
        return table, len(starts) - 1

    # ------------------------------------------------------------------------------------------------ launches
    def _layer(self, dense, act_name, alpha=0.05):
        act = 0 if act_name == "softmax" else _ffi.activation_code(act_name)
        return _ffi.GcnLayerDesc(_ffi.ptr(dense.kernel).value, None if dense.bias is None else dense.bias.data_ptr(),
                                 int(dense.kernel.shape[1]), act, alpha)

    def _descs(self, out):
        """The 1 + depth descriptors of a forward that ends in ``out``."""
        r = self.route
        descs = []
        d = _ffi.GcnTileDesc()
        d.N, d.x, d.K = self.N, self.node.values.data_ptr(), self.K
        d.W_in, d.b_in = r.dense0.kernel.data_ptr(), None if r.dense0.bias is None else r.dense0.bias.data_ptr()
        d.units_in, d.n_layers = r.units, 1
        d.layer[0] = self._layer(r.gcns[0].lay_dense, "linear")
        d.out = self.h[0].data_ptr()
        descs.append(d)
        depth = len(r.gcns)
        for i in range(depth):
            d = _ffi.GcnTileDesc()
            d.N, d.M = self.N, self.M
            d.h, d.ptr = self.h[i % 2].data_ptr(), self.ptr.data_ptr()
            d.perm = None if self.perm is None else self.perm.data_ptr()
            d.send, d.weight = (self.send.data_ptr() if self.M > 0 else None), self.weight.data_ptr() if self.M > 0 else None
            d.agg_act, d.agg_alpha = _ffi.activation_code(r.gcns[i].lay_act.activation), 0.05
            if self.tile_start is not None:
                d.tile_start, d.n_tiles = self.tile_start.data_ptr(), self.n_tiles
            d.units_in = r.units
            if i + 1 < depth:
                d.n_layers = 1
                d.layer[0] = self._layer(r.gcns[i + 1].lay_dense, "linear")
                d.out = self.h[(i + 1) % 2].data_ptr()
            else:
                mlp = r.out_mlp
                d.n_layers = len(mlp.mlp_dense_layer_list)
                for k, dense in enumerate(mlp.mlp_dense_layer_list):
                    d.layer[k] = self._layer(dense, mlp.mlp_activation_layer_list[k].activation)
                d.softmax_last = 1 if mlp.mlp_activation_layer_list[-1].activation == "softmax" else 0
                d.out = out.data_ptr()
            descs.append(d)
        return descs

    def _launch_all(self, descs):
        for d in descs:
            _ffi.call("mp_gcn_tile_f32", ctypes.byref(d), _ffi.stream())

    def _capture(self, out):
        """Capture the launches on the slot's private stream (a captured graph can be launched on any stream)."""
        descs = self._descs(out)
        torch.cuda.current_stream().synchronize()
        exe = ctypes.c_void_p()
        with torch.cuda.stream(self.stream):
            self._launch_all(descs)      # warm-up outside the capture
            self.stream.synchronize()
            _ffi.call("mp_graph_begin", _ffi.stream())
            try:
                self._launch_all(descs)
            finally:
                _ffi.call("mp_graph_end", _ffi.stream(), ctypes.byref(exe))
        return exe

    def _new_out(self):
        return torch.empty((self.N, self.out_units), dtype=torch.float32, device=self.h[0].device)

    def run(self, how):
        """One forward on torch's current stream; returns an (N, out_units) tensor nobody else holds
        (``result_ring.ResultRing``)."""
        self.calls += 1
        if how != "graph":
            out = self._new_out()
            self._launch_all(self._descs(out))
            return out
        got = self._ring.acquire(lambda: (self._new_out(),), lambda bufs: self._capture(bufs[0]))
        if got is None:                  # every result buffer is still held by the caller: static buffer + copy
            if self._static is None:
                self._static = [self._new_out(), None]
                self._static[1] = self._capture(self._static[0])
            _ffi.call("mp_graph_launch", self._static[1], _ffi.stream())
            return self._static[0].clone()
        (out,), graph = got
        _ffi.call("mp_graph_launch", graph, _ffi.stream())
        return out

    def __del__(self):
        try:
            self._ring.destroy()
            if self._static is not None and self._static[1] is not None:
                _ffi.call("mp_graph_destroy", self._static[1])
        except Exception:
            pass


class GcnFusedRoute:
    """The fused forward behind ``GCN.make_model(...)(inputs)`` (node output).  ``mode``: ``auto`` = direct launches on
    the first sight of an input set, graph replay afterwards | ``graph`` | ``eager``."""

    def __init__(self, dense0, gcns, out_mlp, cast, max_slots=4):
        self.dense0, self.gcns, self.out_mlp, self.cast = dense0, gcns, out_mlp, cast
        self.units = int(gcns[0].units)
        self.out_units = int(out_mlp.mlp_dense_layer_list[-1].units)
        self.max_slots = int(max_slots)
        self.mode = "auto"
        self.last = None
        self._slots = {}
        self._wkey = None

    def _weights(self):
        ws = [self.dense0.kernel, self.dense0.bias]
        for g in self.gcns:
            ws += [g.lay_dense.kernel, g.lay_dense.bias]
        for d in self.out_mlp.mlp_dense_layer_list:
            ws += [d.kernel, d.bias]
        return ws

    @staticmethod
    def accepts(inputs):
        from .autograd import needs_grad
        from .ragged import RaggedTensor
        if not (isinstance(inputs, (list, tuple)) and len(inputs) == 3
                and all(isinstance(x, RaggedTensor) for x in inputs)):
            return False
        x, w, idx = (t.values for t in inputs)
        return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
                and w.dtype == torch.float32 and w.dim() == 2 and int(w.shape[1]) == 1 and w.is_contiguous()
                and idx.dtype == torch.int64 and idx.dim() == 2 and int(idx.shape[1]) == 2
                and int(w.shape[0]) == int(idx.shape[0]) and inputs[0].nrows() == inputs[2].nrows()
                and not needs_grad(x, w))

    @staticmethod
    def _key(node, edge_w, idx):
        return (node.values.data_ptr(), edge_w.values.data_ptr(), idx.values.data_ptr(), node.row_splits.data_ptr(),
                idx.row_splits.data_ptr(), tuple(node.values.shape), int(idx.values.shape[0]), node.nrows(),
                idx.values._version, idx.row_splits._version, node.row_splits._version)

    def __call__(self, inputs):
        node, edge_w, idx = inputs
        if int(node.values.shape[1]) != int(self.dense0.kernel.shape[0]):
            raise ValueError("node_attributes have %d features, the model was built for %d"
                             % (int(node.values.shape[1]), int(self.dense0.kernel.shape[0])))
        # the kernels read the weight tensors in place; captured graphs hold their addresses
        wkey = tuple(None if t is None else t.data_ptr() for t in self._weights())
        if wkey != self._wkey:
            self._slots.clear()
            self._wkey = wkey
        key = self._key(node, edge_w, idx)
        slot = self._slots.get(key)
        if slot is None:
            slot = FusedGcn(self, node, edge_w, idx)
            while len(self._slots) >= self.max_slots:
                self._slots.pop(next(iter(self._slots)))
            self._slots[key] = slot
        elif next(reversed(self._slots)) != key:
            self._slots[key] = self._slots.pop(key)
        how = self.mode
        if how == "auto":
            how = "eager" if slot.calls == 0 else "graph"
        self.last = how
        out = node.with_values(slot.run(how))
        if self.cast is None:
            return out
        if node.nrows() == 1:            # one graph: the padded tensor is the value matrix itself
            return out.values.view(1, slot.N, slot.out_units)
        return self.cast(out)

    def slot_of(self, inputs):
        return self._slots.get(self._key(*inputs))

    def release(self):
        torch.cuda.synchronize()
        self._slots.clear()
