"""Fused SchNet forward: the arithmetic of kgcnn/literature/Schnet.py:104-148 in eight kernels, replayed from a HIP graph.

    stage0              edge_prepare (index shift, receiver/sender split, flags, distance) and node_in
                        (Embedding -> Dense(64->128) -> Dense_nobias) on disjoint workgroups             (1 launch)
    per block:          cfconv_gauss_fused (Gauss basis, filter MLP, gather, multiply, segment-sum)     (depth launches)
                        node_update / node_last (2-3 chained Dense on the node tile, residual)         (depth launches)
    readout             PoolingNodes(sum) + output MLP                                                 (1 launch)

Every buffer is allocated when a batch is bound; ``forward()`` only replays the captured graph.
"""
import ctypes
import os

import numpy as np
import torch

from . import _ffi
from .layers.base import weight_epoch
from .result_ring import ResultRing


_SSP = ("kgcnn>shifted_softplus", "shifted_softplus")
_SUM = ("sum", "segment_sum", "reduce_sum")


def _as_list(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def head_of(config):
    """Which readout a configuration ends in: ``"mlp"`` = the reference default (last_mlp [128, 64], PoolingNodes(sum),
    output_mlp [64, 1] with (ssp, linear)); ``"linear"`` = the fork's force configuration (force_schnet.py:139-156:
    last_mlp [128, 64, 1] with (ssp, ssp, linear), PoolingNodes(sum), no output MLP); ``None`` = neither."""
    lm = config["last_mlp"]
    units = _as_list(lm["units"], 1)
    acts = _as_list(lm.get("activation"), len(units))
    if not all(_as_list(lm.get("use_bias", True), len(units))):
        return None
    if config.get("use_output_mlp", True):
        om = config["output_mlp"]
        if (units == [128, 64] and all(a in _SSP for a in acts) and _as_list(om["units"], 1) == [64, 1]
                and _as_list(om.get("activation"), 2)[0] in _SSP and _as_list(om.get("activation"), 2)[1] in ("linear", None)
                and all(_as_list(om.get("use_bias", True), 2))):
            return "mlp"
        return None
    if units == [128, 64, 1] and all(a in _SSP for a in acts[:2]) and acts[2] in ("linear", None):
        return "linear"
    return None


def supports(config):
    """True if a ``Schnet.make_model`` configuration (the merged keyword dictionary) maps onto the fused kernels:
    node numbers (float32 or int64) through a 64- or 128-wide embedding, distances and Gauss basis made inside the
    model, 128 units with shifted softplus and sum pooling, graph output through one of the two heads of ``head_of``.
    Anything else runs the layer path (kgcnn/literature/Schnet.py:104-148 op by op)."""
    try:
        ia = config["interaction_args"]
        inputs = config.get("inputs")
        if inputs is not None and (len(inputs[0]["shape"]) != 1 or tuple(inputs[1]["shape"])[-1] != 3):
            return False
        return bool(
            config.get("make_distance", True) and config.get("expand_distance", True)
            and config.get("output_embedding", "graph") == "graph"
            and ia.get("units") == 128 and ia.get("cfconv_pool", "sum") in _SUM
            and ia.get("activation", _SSP[0]) in _SSP and ia.get("use_bias", True) is True
            and config["input_embedding"]["node"]["output_dim"] in (64, 128)
            and head_of(config) is not None
            and config["node_pooling_args"].get("pooling_method") in _SUM
            and 1 <= int(config["gauss_args"]["bins"]) <= 32 and float(config["gauss_args"]["sigma"]) != 0.0
            and int(config["depth"]) >= 1)
    except (KeyError, TypeError, IndexError):
        return False


def node_weight_names(depth):
    """Keras kernels the node-side kernels read as ``mp_schnet_node_pack_f32`` images."""
    names = ["dense0/kernel", "last_mlp/0/kernel", "last_mlp/1/kernel"]
    for i in range(depth):
        names += ["interaction%d/dense%d/kernel" % (i, k) for k in (1, 2, 3)]
    return names


def pack_weights(p, depth, bins, out=None):
    """Kernel-side images of the weights: ``mp_cfconv_pack_f32`` (filter MLP of every interaction block in the cfconv
    kernel's LDS order) and ``mp_schnet_node_pack_f32`` (node-side matrices in register-slice order).  Packed once per
    weight update; ``out`` re-fills existing images in place (captured graphs keep pointing at them)."""
    nfl = _ffi.lib().mp_cfconv_packed_floats()
    if out is None:
        out = {"cfconv": [torch.empty(nfl, dtype=torch.float32, device="cuda") for _ in range(depth)],
               "node": {k: torch.empty(p[k].numel(), dtype=torch.float32, device="cuda")
                        for k in node_weight_names(depth)},
               # the same matrices as three bf16 pieces per element (the forward's node kernels on the bf16 pipe)
               "node_bf": {k: torch.empty(p[k].numel() * 3 // 2, dtype=torch.float32, device="cuda")
                           for k in node_weight_names(depth)}}
    for i in range(depth):
        pre = "interaction%d/cfconv/" % i
        _ffi.call("mp_cfconv_pack_f32", _ffi.ptr(p[pre + "dense1/kernel"]), _ffi.ptr(p.get(pre + "dense1/bias")),
                  int(bins), _ffi.ptr(p[pre + "dense2/kernel"]), _ffi.ptr(p.get(pre + "dense2/bias")),
                  _ffi.ptr(out["cfconv"][i]), _ffi.stream())
    for k, image in out["node"].items():
        _ffi.call("mp_schnet_node_pack_f32", _ffi.ptr(p[k]), int(p[k].shape[0]), int(p[k].shape[1]), _ffi.ptr(image),
                  _ffi.stream())
    for k, image in out.get("node_bf", {}).items():
        _ffi.call("mp_schnet_node_pack_bf16_f32", _ffi.ptr(p[k]), int(p[k].shape[0]), int(p[k].shape[1]),
                  _ffi.ptr(image), _ffi.stream())
    torch.cuda.current_stream().synchronize()
    return out


def _round_up(x, unit):
    return ((max(int(x), 1) + unit - 1) // unit) * unit


class WorkSet:
    """Work buffers of one batch slot at a bucketed capacity, reusable by later batches of similar size, plus the
    direct-launch descriptor whose weight and work-buffer fields are filled once.  ``agg`` is zero between forwards (the
    node kernels re-zero the rows they consume, rows beyond a batch's N are never written), ``flags`` is zero unless a
    kernel flagged an error, so a recycled set needs no fill launch."""
    __slots__ = ("n_cap", "m_cap", "g_cap", "recv", "send", "dist", "flags", "n", "x", "agg", "h", "desc", "stream_key",
                 "_union")

    def __init__(self, n_cap, m_cap, g_cap, stream_key):
        dev = "cuda"
        self.n_cap, self.m_cap, self.g_cap, self.stream_key = n_cap, m_cap, g_cap, stream_key
        self.recv = torch.empty(m_cap, dtype=torch.int32, device=dev)
        self.send = torch.empty(m_cap, dtype=torch.int32, device=dev)
        self.dist = torch.empty(m_cap, dtype=torch.float32, device=dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self.n = torch.empty((n_cap, 128), dtype=torch.float32, device=dev)
        self.x = torch.empty((n_cap, 128), dtype=torch.float32, device=dev)
        self.agg = torch.zeros((n_cap, 128), dtype=torch.float32, device=dev)
        self.h = torch.empty((n_cap, 64), dtype=torch.float32, device=dev)
        self.desc = None
        self._union = {}

    def union(self, z_dtype):
        """Input tensors of a launch group's union batch at this set's capacity (node numbers of ``z_dtype``, coordinates,
        sample indices, both row splits) and the ``mp_concat_batches`` descriptor whose destination fields point at them:
        made on first use, recycled with the set (a group of never-seen batches then allocates nothing)."""
        got = self._union.get(z_dtype)
        if got is None:
            dev = "cuda"
            z = torch.empty(self.n_cap, dtype=z_dtype, device=dev)
            xyz = torch.empty((self.n_cap, 3), dtype=torch.float32, device=dev)
            idx = torch.empty((self.m_cap, 2), dtype=torch.int64, device=dev)
            ns = torch.empty(self.g_cap + 1, dtype=torch.int64, device=dev)
            es = torch.empty(self.g_cap + 1, dtype=torch.int64, device=dev)
            d = _ffi.ConcatDesc()
            d.z_is_i64 = 1 if z_dtype == torch.int64 else 0
            d.z, d.xyz, d.idx = z.data_ptr(), xyz.data_ptr(), idx.data_ptr()
            d.node_splits, d.edge_splits = ns.data_ptr(), es.data_ptr()
            got = self._union[z_dtype] = (z, xyz, idx, ns, es, d, ctypes.byref(d))
        return got


class WorkArena:
    """Free work sets of a route, per (stream, size bucket).  A batch that is seen once (``model.predict`` over a dataset,
    kgcnn/data/base.py:203-239) binds, launches and is dropped; its ~10 work tensors then cost ~30 us of allocator calls
    and three fill launches per batch - more host time than the forward's eight launches.  Sets are handed back when their
    slot object dies and re-issued only to binds under the SAME stream (stream order is what makes the reuse safe: the
    previous owner's kernels were queued on that stream before the next owner's)."""
    N_UNIT, M_UNIT, G_UNIT = 512, 8192, 64

    def __init__(self, max_free=32):
        self.free = {}
        self.max_free = max_free
        self.taken = self.made = 0

    @staticmethod
    def _bucket(x, unit):
        # capacity classes an eighth of the size's power of two apart (never finer than ``unit``): batches of a dataset
        # differ by a few per cent in N and M, launch groups of them by the same few per cent of a five times larger
        # size - classes of a fixed width would give nearly every group a class of its own and the arena no hits
        x = max(int(x), 1)
        return _round_up(x, max(unit, (1 << (x.bit_length() - 1)) >> 3))

    def take(self, stream_key, n, m, g):
        key = (stream_key, self._bucket(n, self.N_UNIT), self._bucket(m, self.M_UNIT), self._bucket(g, self.G_UNIT))
        bucket = self.free.get(key)
        if bucket:
            self.taken += 1
            return bucket.pop()
        self.made += 1
        return WorkSet(key[1], key[2], key[3], stream_key)

    def give(self, ws):
        bucket = self.free.setdefault((ws.stream_key, ws.n_cap, ws.m_cap, ws.g_cap), [])
        if len(bucket) < self.max_free:
            bucket.append(ws)

    def clear(self):
        self.free.clear()


class FusedSchnet:
    """One batch slot of the fused forward: work buffers, a HIP stream and the captured graph of ONE bound batch.

    ``params`` maps the names of ``gcnn_keras_amd.synth.schnet_params`` to NumPy arrays (copied to HBM) or to device
    tensors (used in place: the slots of a ``Schnet.make_model`` model read the model's own weight tensors).
    ``packed`` optionally shares the weight images (``pack_weights``) of another slot of the same model."""

    def __init__(self, params, depth=3, gauss_args=None, fast_softplus=True, use_graph=True, cfconv_flags=0,
                 packed=None):
        if not _ffi.has_gpu():
            raise _ffi.EngineError("FusedSchnet needs an MI355X (no CPU fallback)")
        self.depth = depth
        self.gauss = dict(gauss_args or {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4})
        self.flags_arg = (1 if fast_softplus else 0) | 2 | int(cfconv_flags)   # bit 1: node weights as packed images
        self._stream_ptr = None
        self._desc = self._desc_ref = None
        self.use_graph = use_graph
        if packed is not None:      # a route's slot: ``params`` is the route's dictionary of live device tensors
            self.p = params
        else:
            self.p = {k: (v if torch.is_tensor(v) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda())
                      for k, v in params.items() if v is not None}
        self.emb_dim = int(self.p["embedding"].shape[1])
        if self.emb_dim not in (64, 128) or tuple(self.p["dense0/kernel"].shape) != (self.emb_dim, 128):
            raise ValueError("FusedSchnet is built for embedding width 64 / 128 and 128 units")
        # head: output MLP [64, 1] after the pooling (reference default) or last_mlp's own third layer Dense(1, linear)
        # before it (use_output_mlp=False: the fork's force_schnet.py configuration)
        self.linear_head = "output_mlp/0/kernel" not in self.p
        # filter-MLP weights of every block in the cfconv kernel's LDS image order (packed once per weight update)
        images = packed if packed is not None else pack_weights(self.p, depth, int(self.gauss["bins"]))
        self.packed, self.node_images = images["cfconv"], images["node"]
        # flags bit 6: node-side GEMMs on the bf16 matrix pipe (exact FP32 emulation, csrc/mp_node_tile.h) from the
        # bf16-piece images; MPENGINE_NODE_BF16=0 keeps the FP32 matrix instructions
        if "node_bf" in images and os.environ.get("MPENGINE_NODE_BF16", "1") != "0":
            self.node_images = images["node_bf"]
            self.flags_arg |= 64
        self._own_stream = None     # made on first use (capture / engine.SchnetForward): a slot that is bound, launched
        self.graph = None           # once and dropped never needs one
        self._ring = None
        self.num_launches = 1 + 2 * depth + 1
        self._b = None
        self._work = self._arena = None
        self._foreign_stream = False
        self._pre = None            # launch group: the concatenation of the member batches runs in front of every forward

    @property
    def stream(self):
        if self._own_stream is None:
            self._own_stream = torch.cuda.Stream()
        return self._own_stream

    @stream.setter
    def stream(self, value):
        self._own_stream = value

    # ------------------------------------------------------------------------------------------------ binding
    def bind(self, b, n, m, g, known_flags=None, arena=None, work=None):
        """Attach a resident batch (dict of device tensors: z, xyz, idx, ns, es + host node splits) and take its work
        buffers - from ``arena`` (a route's ``WorkArena``: a recycled set of the size bucket, nothing allocated, nothing
        filled) or freshly allocated on the current stream.  ``known_flags``: the MP_FLAG_* word of the index list if a
        producer (host packer, on-GPU SetRange) already established it - otherwise one index pass runs and its flag word is
        read back (the only host synchronisation of a batch's life, on the current stream only)."""
        self._b, self.N, self.M, self.G = b, n, m, g
        i64 = 256 if b["z"].dtype == torch.int64 else 0      # flags bit 8: int64 node numbers (the fork's input dtype)
        self.node_flags = (self.flags_arg & (3 | 64 | 512)) | i64   # bit 9: node chains on half the CUs (launches in flight)
        dev = "cuda"
        if arena is not None:
            ws = work if work is not None else arena.take(_ffi.stream_handle(), n, m, g)   # (work: taken by the caller)
            self._work, self._arena = ws, arena
            self.recv, self.send, self.dist, self.flags = ws.recv, ws.send, ws.dist, ws.flags
            self.n, self.x, self.agg, self.h = ws.n, ws.x, ws.agg, ws.h
            self.out = torch.empty((g, 1), dtype=torch.float32, device=dev)   # the readout writes every row
        else:
            self.recv = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
            self.send = torch.empty(max(m, 1), dtype=torch.int32, device=dev)
            self.dist = torch.empty(max(m, 1), dtype=torch.float32, device=dev)
            self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
            self.n = torch.empty((n, 128), dtype=torch.float32, device=dev)
            self.x = torch.empty((n, 128), dtype=torch.float32, device=dev)
            self.agg = torch.zeros((n, 128), dtype=torch.float32, device=dev)
            self.h = torch.empty((n, 64), dtype=torch.float32, device=dev)
            self.out = torch.zeros((g, 1), dtype=torch.float32, device=dev)
        self.perm = self.recv_sorted = self.sort_ws = None
        # sortedness of the receiver column is a property of the batch: decide once, outside the timed region
        if known_flags is None:
            self._prepare()
            f = int(self.flags.item())
            self.flags.zero_()
        else:
            f = int(known_flags)
        if f & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")
        self.sorted = not (f & _ffi.MP_FLAG_UNSORTED_COL0)
        if not self.sorted and m > 0:
            nbytes = ctypes.c_size_t(0)
            _ffi.call("mp_sort_workspace_bytes", m, ctypes.byref(nbytes))
            self.sort_ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
            self.sort_ws_bytes = nbytes.value
            self.recv_sorted = torch.empty(m, dtype=torch.int32, device=dev)
            self.perm = torch.empty(m, dtype=torch.int32, device=dev)
        splits = np.asarray(b["ns_host"])
        rows = g
        while rows > 0 and splits[rows] == splits[rows - 1]:
            rows -= 1
        self.out_rows = rows  # tf.math.segment_sum drops trailing empty graphs (kgcnn/layers/pooling.py:215-219)
        self._drop_graphs()
        self._desc = self._desc_ref = None

    def _prepare(self):
        b = self._b
        _ffi.call("mp_edge_prepare_i64_f32", _ffi.ptr(b["idx"]), self.M, _ffi.ptr(b["ns"]), _ffi.ptr(b["es"]), self.G,
                  self.N, _ffi.ptr(b["xyz"]), _ffi.ptr(self.recv), _ffi.ptr(self.send), _ffi.ptr(self.dist),
                  _ffi.ptr(self.flags), _ffi.stream())

    def _cfconv(self, i, out):
        ga = self.gauss
        recv = self.recv if self.sorted else self.recv_sorted
        _ffi.call("mp_cfconv_gauss_fused_f32", _ffi.ptr(self.x), self.N, _ffi.ptr(self.dist), int(ga["bins"]),
                  float(ga["distance"]), float(ga["sigma"]), float(ga["offset"]), _ffi.ptr(self.packed[i]),
                  _ffi.ptr(recv), _ffi.ptr(self.send), _ffi.ptr(self.perm), self.M, self.flags_arg, _ffi.ptr(out),
                  _ffi.stream())

    def _launch_all(self, out=None):
        # One linear chain on one stream.  (A two-branch graph - node_in beside edge_prepare - was measured 9 % SLOWER
        # at config 2: the fork/join costs more than the ~5 us of overlap it buys.)
        p, b, w = self.p, self._b, self.node_images
        if self._pre is not None:
            self._pre()
        if self.sorted or self.M == 0:
            # stage 0: node-input chain and edge preparation in one launch (independent work on disjoint workgroups)
            _ffi.call("mp_schnet_stage0_f32", _ffi.ptr(b["z"]), self.N, _ffi.ptr(p["embedding"]),
                      int(p["embedding"].shape[0]), self.emb_dim, _ffi.ptr(w["dense0/kernel"]), _ffi.ptr(p.get("dense0/bias")),
                      _ffi.ptr(w["interaction0/dense1/kernel"]), _ffi.ptr(self.n), _ffi.ptr(self.x),
                      _ffi.ptr(b["idx"]), self.M, _ffi.ptr(b["ns"]), _ffi.ptr(b["es"]), self.G, _ffi.ptr(b["xyz"]),
                      _ffi.ptr(self.recv), _ffi.ptr(self.send), _ffi.ptr(self.dist), _ffi.ptr(self.flags),
                      self.node_flags, _ffi.stream())
        else:
            self._prepare()
            _ffi.call("mp_sort_segments_i32", _ffi.ptr(self.recv), self.M, _ffi.ptr(self.recv_sorted),
                      _ffi.ptr(self.perm), _ffi.ptr(self.sort_ws), self.sort_ws_bytes, _ffi.stream())
            _ffi.call("mp_schnet_node_in_f32", _ffi.ptr(b["z"]), self.N, _ffi.ptr(p["embedding"]),
                      int(p["embedding"].shape[0]), self.emb_dim, _ffi.ptr(w["dense0/kernel"]), _ffi.ptr(p.get("dense0/bias")),
                      _ffi.ptr(w["interaction0/dense1/kernel"]), _ffi.ptr(self.n), _ffi.ptr(self.x),
                      self.node_flags, _ffi.stream())
        for i in range(self.depth):
            pre = "interaction%d/" % i
            self._cfconv(i, self.agg)
            if i + 1 < self.depth:
                _ffi.call("mp_schnet_node_update_f32", _ffi.ptr(self.agg), self.N, _ffi.ptr(w[pre + "dense2/kernel"]),
                          _ffi.ptr(p.get(pre + "dense2/bias")), _ffi.ptr(w[pre + "dense3/kernel"]),
                          _ffi.ptr(p.get(pre + "dense3/bias")), _ffi.ptr(self.n),
                          _ffi.ptr(w["interaction%d/dense1/kernel" % (i + 1)]), _ffi.ptr(self.x), self.node_flags,
                          _ffi.stream())
            else:
                _ffi.call("mp_schnet_node_last_f32", _ffi.ptr(self.agg), self.N, _ffi.ptr(w[pre + "dense2/kernel"]),
                          _ffi.ptr(p.get(pre + "dense2/bias")), _ffi.ptr(w[pre + "dense3/kernel"]),
                          _ffi.ptr(p.get(pre + "dense3/bias")), _ffi.ptr(self.n), _ffi.ptr(w["last_mlp/0/kernel"]),
                          _ffi.ptr(p.get("last_mlp/0/bias")), _ffi.ptr(w["last_mlp/1/kernel"]),
                          _ffi.ptr(p.get("last_mlp/1/bias")), _ffi.ptr(self.h), self.node_flags, _ffi.stream())
        wo0, bo0, wo1, bo1 = self._head()
        _ffi.call("mp_schnet_readout_f32", _ffi.ptr(self.h), _ffi.ptr(b["ns"]), self.G, _ffi.ptr(wo0), _ffi.ptr(bo0),
                  _ffi.ptr(wo1), _ffi.ptr(bo1), _ffi.ptr(self.out if out is None else out), _ffi.stream())

    def _head(self):
        p = self.p
        if self.linear_head:
            return None, None, p["last_mlp/2/kernel"], p.get("last_mlp/2/bias")
        return (p["output_mlp/0/kernel"], p.get("output_mlp/0/bias"), p["output_mlp/1/kernel"],
                p.get("output_mlp/1/bias"))

    # ------------------------------------------------------------------------------------------------ current stream
    def _capture(self, out=None):
        """Capture the eight launches on this slot's private stream (a captured graph can be launched on any stream);
        ``out``: the result buffer the readout of THIS graph writes (default: the slot's static buffer)."""
        with torch.cuda.stream(self.stream):
            self._launch_all(out)  # warm-up outside capture (lazy module load, function attributes)
            self.stream.synchronize()
            _ffi.call("mp_graph_begin", _ffi.stream())
            try:
                self._launch_all(out)
            finally:
                exe = ctypes.c_void_p()
                _ffi.call("mp_graph_end", _ffi.stream(), ctypes.byref(exe))
        if out is None:
            self.graph = exe
        return exe

    # ------------------------------------------------------------------------------------------------ result ring
    def run_graph_fresh(self):
        """Replay the forward on torch's current stream into a result buffer nobody else references
        (``result_ring.ResultRing``: own captured graph per buffer, no copy launch); returns that ``(G', 1)`` tensor, or
        ``None`` when every ring buffer is still held by a caller."""
        if self._ring is None:
            self._ring = ResultRing()
        got = self._ring.acquire(lambda: (torch.zeros((self.G, 1), dtype=torch.float32, device="cuda"),),
                                 lambda bufs: self._capture_synced(bufs[0]))
        if got is None:
            return None
        (out,), graph = got
        st = _ffi.stream()
        self._note_stream(st)
        _ffi.call("mp_graph_launch", graph, st)
        return out if self.out_rows == self.G else out[:self.out_rows]

    def _capture_synced(self, out):
        torch.cuda.current_stream().synchronize()
        return self._capture(out)

    def _drop_graphs(self):
        if getattr(self, "graph", None) is not None:
            try:
                _ffi.call("mp_graph_destroy", self.graph)
            except Exception:
                pass
        if getattr(self, "_ring", None) is not None:
            self._ring.destroy()
        self.graph = None
        self._ring = None

    def _note_stream(self, st):
        """A recycled work set goes back to the arena of the stream it was taken under; a slot that was also launched on
        another stream is not recycled (its last kernels are not ordered before that stream's next bind)."""
        if self._work is not None and (st.value or 0) != self._work.stream_key:   # c_void_p(0).value is None
            self._foreign_stream = True

    def hand_out_static(self):
        """The static result buffer itself as the caller's tensor (first, direct call of an arena-bound batch: no copy
        launch).  The slot makes a new static buffer if it ever needs one again."""
        self._out_handed = True
        return self.out if self.out_rows == self.G else self.out[:self.out_rows]

    def _fresh_static(self):
        if getattr(self, "_out_handed", False):
            self._out_handed = False
            self.out = torch.zeros((self.G, 1), dtype=torch.float32, device="cuda")
            if self._desc is not None:
                self._desc.out = self.out.data_ptr()
            if self.graph is not None:        # captured against the buffer the caller now owns
                _ffi.call("mp_graph_destroy", self.graph)
                self.graph = None

    def run_current(self, how="graph"):
        """One forward of the bound batch on torch's CURRENT stream (ordinary stream semantics for the caller):
        ``graph`` replays the captured forward (captured on first use), ``direct`` issues the eight launches from one
        C-ABI call (``mp_schnet_forward_launch``), ``eager`` issues them one engine call each.  Returns this slot's
        static ``(G', 1)`` output buffer."""
        self._fresh_static()
        st = _ffi.stream()
        self._note_stream(st)
        if how == "graph":
            if self.graph is None:
                torch.cuda.current_stream().synchronize()
                self._capture()
            _ffi.call("mp_graph_launch", self.graph, _ffi.stream())
        elif how == "direct" and (self.sorted or self.M == 0) and self.depth <= _ffi.MP_SCHNET_MAX_DEPTH:
            if self._desc is None:
                self._desc = self._descriptor()
            if self._pre is not None:
                self._pre()
            _ffi.call("mp_schnet_forward_launch", ctypes.byref(self._desc), _ffi.stream())
        else:
            self._launch_all()
        return self.out if self.out_rows == self.G else self.out[:self.out_rows]

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self):
        """One forward of the bound batch; returns the (G', 1) prediction tensor (valid after stream sync)."""
        cur = torch.cuda.current_stream()
        same = cur == self.stream  # callers that already run on the engine's stream pay no cross-stream events
        if not same:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if not self.use_graph:
                self._launch_all()
            else:
                if self.graph is None:
                    self._launch_all()  # warm-up outside capture (lazy module load, function attributes)
                    torch.cuda.synchronize()
                    _ffi.call("mp_graph_begin", _ffi.stream())
                    try:
                        self._launch_all()
                    finally:
                        exe = ctypes.c_void_p()
                        _ffi.call("mp_graph_end", _ffi.stream(), ctypes.byref(exe))
                    self.graph = exe
                _ffi.call("mp_graph_launch", self.graph, _ffi.stream())
        if not same:
            cur.wait_stream(self.stream)
        return self.out if self.out_rows == self.G else self.out[:self.out_rows]

    def replay(self):
        """Launch the captured forward on this slot's own stream with no Python stream bookkeeping (one C-ABI call):
        what a serving loop calls per batch once the batch is bound.  ``forward()`` must have run once (capture)."""
        if self.graph is None:
            self.forward()
            return self.out
        if self._stream_ptr is None:
            self._stream_ptr = ctypes.c_void_p(self.stream.cuda_stream)
            self._launch = _ffi.lib().mp_graph_launch
        _ffi.check(self._launch(self.graph, self._stream_ptr))
        return self.out if self.out_rows == self.G else self.out[:self.out_rows]

    # ------------------------------------------------------------------------------------------------ direct launch
    def _descriptor(self):
        """``mp_schnet_forward_desc`` of the bound batch (receiver-sorted batches only)."""
        p, b, ga, w = self.p, self._b, self.gauss, self.node_images
        addr = lambda t: None if t is None else t.data_ptr()
        ws = self._work
        if ws is not None and ws.desc is not None:   # a recycled set: only the batch's own fields change
            d = ws.desc
            d.N, d.M, d.G = self.N, self.M, self.G
            d.flags = self.flags_arg | (self.node_flags & 256)
            d.numbers, d.xyz, d.idx = addr(b["z"]), addr(b["xyz"]), addr(b["idx"])
            d.node_splits, d.edge_splits = addr(b["ns"]), addr(b["es"])
            d.out = addr(self.out)
            return d
        d = _ffi.SchnetForwardDesc()
        if ws is not None:
            ws.desc = d
        d.N, d.M, d.G = self.N, self.M, self.G
        d.depth, d.vocab, d.bins = self.depth, int(p["embedding"].shape[0]), int(ga["bins"])
        d.flags = self.flags_arg | (self.node_flags & 256)
        d.g_distance, d.g_sigma, d.g_offset = float(ga["distance"]), float(ga["sigma"]), float(ga["offset"])
        d.numbers, d.xyz, d.idx = addr(b["z"]), addr(b["xyz"]), addr(b["idx"])
        d.node_splits, d.edge_splits = addr(b["ns"]), addr(b["es"])
        d.embedding, d.W0, d.b0 = addr(p["embedding"]), addr(w["dense0/kernel"]), addr(p.get("dense0/bias"))
        for i in range(self.depth):
            pre = "interaction%d/" % i
            d.Wx[i], d.packed[i] = addr(w[pre + "dense1/kernel"]), addr(self.packed[i])
            d.W2[i], d.b2[i] = addr(w[pre + "dense2/kernel"]), addr(p.get(pre + "dense2/bias"))
            d.W3[i], d.b3[i] = addr(w[pre + "dense3/kernel"]), addr(p.get(pre + "dense3/bias"))
        d.Wl0, d.bl0 = addr(w["last_mlp/0/kernel"]), addr(p.get("last_mlp/0/bias"))
        d.Wl1, d.bl1 = addr(w["last_mlp/1/kernel"]), addr(p.get("last_mlp/1/bias"))
        wo0, bo0, wo1, bo1 = self._head()
        d.Wo0, d.bo0, d.Wo1, d.bo1 = addr(wo0), addr(bo0), addr(wo1), addr(bo1)
        d.emb_dim = self.emb_dim
        d.recv, d.send, d.dist, d.flags_word = addr(self.recv), addr(self.send), addr(self.dist), addr(self.flags)
        d.n, d.x, d.agg, d.h, d.out = addr(self.n), addr(self.x), addr(self.agg), addr(self.h), addr(self.out)
        return d

    def launch_direct(self):
        """The same eight launches as the captured graph, issued directly by ONE C-ABI call on this slot's stream
        (``mp_schnet_forward_launch``, ~22 us of host time): for callers whose batches change shape every call, where
        capturing a graph per batch (milliseconds) cannot pay off.  The call holds no Python lock (ctypes), so slots can
        be driven by one host thread each.  Throughput with batches in flight is the same as with graph replay (measured
        46.6 vs 46.9 us per step) - the GPU, not the submission path, is the limit there."""
        if not (self.sorted or self.M == 0) or self.depth > _ffi.MP_SCHNET_MAX_DEPTH:
            return self.replay()
        if self._desc_ref is None:
            self._desc = self._descriptor()
            self._desc_ref = ctypes.byref(self._desc)
            self._direct = _ffi.lib().mp_schnet_forward_launch
            self._stream_ptr = ctypes.c_void_p(self.stream.cuda_stream)
        elif self._stream_ptr is None:
            self._stream_ptr = ctypes.c_void_p(self.stream.cuda_stream)
        _ffi.check(self._direct(self._desc_ref, self._stream_ptr))
        return self.out if self.out_rows == self.G else self.out[:self.out_rows]

    def check_flags(self):
        torch.cuda.synchronize()
        f = int(self.flags.item())
        if f and self._work is not None:
            self.flags.zero_()          # the word belongs to a recycled work set: the next owner starts clean
        if f & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")
        if self.sorted and (f & _ffi.MP_FLAG_UNSORTED_COL0):
            raise _ffi.EngineError("batch was bound as receiver-sorted but an unsorted index list was replayed")

    # ------------------------------------------------------------------------------------------------ roofline
    def roofline(self, hbm_peak_gbs, mfma_peak_tf, iters=50):
        """The cfconv kernel (>= 70 % of the forward's flops), timed alone with HIP events on the stream it runs on."""
        from .engine import _HipTimer
        scratch = torch.zeros_like(self.agg)
        bins = int(self.gauss["bins"])
        # `iters` back-to-back launches captured in a HIP graph and replayed between two events: issued one by one through
        # ctypes the launches arrive every ~10-11 us - the Python call rate, not the kernel (rocprofv3's 10.0 us average
        # of the same kernel inside the forward is the cross-check, profiles/r02_fused_config2_kernel_stats.csv)
        with torch.cuda.stream(self.stream):
            self._cfconv(0, scratch)
            torch.cuda.synchronize()
            _ffi.call("mp_graph_begin", _ffi.stream())
            exe = ctypes.c_void_p()
            try:
                for _ in range(iters):
                    self._cfconv(0, scratch)
            finally:
                _ffi.call("mp_graph_end", _ffi.stream(), ctypes.byref(exe))
            try:
                ms = _HipTimer().time_ms(lambda: _ffi.call("mp_graph_launch", exe, _ffi.stream()), 4) / iters
            finally:
                _ffi.call("mp_graph_destroy", exe)
        torch.cuda.synchronize()
        flops = float(self.M) * (2.0 * (bins * 128 + 128 * 128) + 2.0 * 128)
        # algorithmic bytes of one launch: distance + ids per edge, sender rows once, output rows once
        alg_bytes = float(self.M) * (4 + 8) + 2.0 * self.N * 128 * 4
        achieved = flops / (ms * 1e-3) / 1e12
        # matrix-pipe occupancy of the instruction mix actually issued, per 32-edge tile: GEMM2 = 8 k blocks x 4 column
        # blocks x 6 v_mfma_f32_32x32x16_bf16 (32 cycles each; FP32 emulated exactly from three bf16 pieces per operand);
        # GEMM1 = 2 k blocks x 4 hidden blocks x 6 of the same (basis + bias row <= 32 slots: 20 bins), else
        # ceil((bins+1)/2) k steps x 4 blocks of v_mfma_f32_32x32x2_f32 (64 cycles each)
        tiles = (self.M + 31) // 32
        gemm1_cycles = 2 * 4 * 6 * 32 if bins == 20 else ((bins + 2) // 2) * 4 * 64
        mfma_cycles = tiles * (gemm1_cycles + 8 * 4 * 6 * 32)
        pipe_busy = mfma_cycles / (1024 * 2.4e9 * ms * 1e-3)
        # the build cfconv_dispatch (csrc/mp_cfconv.hip) picks for this launch: eight waves per workgroup when the rounds x cost
        # product favours them (from 3072 tiles on), forced by flag bit 2 / 3; fast softplus and 20 / 25 bins only
        waves = 4
        if tiles >= 3072 and 37 * ((tiles + 2047) // 2048) < 20 * ((tiles + 1023) // 1024):
            waves = 8
        waves = 8 if self.flags_arg & 4 else (4 if self.flags_arg & 8 else waves)
        if not (self.flags_arg & 1 and bins in (20, 25)):
            waves = 4
        return {"bound": "mfma", "kernel": "cfconv_fused_kernel<%d,gauss> (SchNetCFconv, one interaction block)" % waves,
                "achieved": achieved, "peak": mfma_peak_tf, "unit": "TFLOP/s", "frac": achieved / mfma_peak_tf,
                "traffic": None, "avg_launch_us": ms * 1e3, "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": alg_bytes,
                "hbm_gbs_at_this_rate": alg_bytes / (ms * 1e-3) / 1e9, "hbm_peak_gbs": hbm_peak_gbs,
                "peak_note": "peak = dense FP32 MFMA peak (MI355X_MICROARCH.md); achieved = algorithmic FP32 flops / "
                             "time.  The filter GEMMs (K = 21 and K = 128) are issued as six bf16 MFMAs per 16 k - "
                             "an exact FP32 emulation from three bf16 pieces per operand, FP32 accumulate, error equal to "
                             "the FP32 MFMA chain's (scripts/probes/bf16x3_probe.hip) - so frac can exceed 1 at large "
                             "batches; matrix_pipe_busy is the share of all 1024 SIMDs' cycles (2.4 GHz) taken by the "
                             "MFMA instructions actually issued",
                "matrix_pipe_busy": pipe_busy}

    def __del__(self):
        try:
            self._drop_graphs()
            if self._work is not None and self._arena is not None and not self._foreign_stream:
                self._arena.give(self._work)     # every launch of this slot ran on the stream the set belongs to
            self._work = None
        except Exception:
            pass


class SchnetGroup:
    """A launch group: k bound batches served by ONE launch sequence.  The members' tensors are concatenated on the device
    (``mp_concat_batches``: one launch, ~0.5 MB per 128-graph member) into this group's own union batch, on which an
    ordinary batch slot runs; results are views of one fresh ``(sum G, 1)`` tensor, cut at the members' own row counts.
    Graphs of a disjoint batch do not interact, so every member gets the rows a forward of its own would give (up to the
    rounding order of the cfconv boundary sums, whose tile boundaries move: ~1e-6 relative)."""

    def __init__(self, route, inputs_list):
        k = len(inputs_list)
        if not 1 <= k <= _ffi.MP_CONCAT_MAX:
            raise ValueError("a launch group holds 1..%d batches" % _ffi.MP_CONCAT_MAX)
        self.members = [tuple(x) for x in inputs_list]          # keeps the member tensors (and their addresses) alive
        z0 = inputs_list[0][0].values
        if any(x[0].values.dtype != z0.dtype for x in inputs_list):
            raise ValueError("the members of a launch group must share the node-number dtype")
        ns_host = [np.asarray(x[0].row_splits_host(), dtype=np.int64) for x in inputs_list]
        sizes = [(int(x[0].values.shape[0]), int(x[2].values.shape[0]), x[0].nrows()) for x in inputs_list]
        n, m, g = (sum(t[i] for t in sizes) for i in range(3))
        # the union's input tensors live in the work set (capacity of the size bucket; the kernels take addresses and the
        # sizes n, m, g): a group of never-seen batches takes a recycled set and allocates nothing
        work = route._arena.take(_ffi.stream_handle(), n, m, g)
        self.z, self.xyz, self.idx, self.ns, self.es, d, self._desc_ref = work.union(z0.dtype)
        d.k = k
        for b, (x, (nb, mb, gb)) in enumerate(zip(inputs_list, sizes)):
            node, xyz, idx = x
            src = d.src[b]
            src.z, src.xyz, src.idx = node.values.data_ptr(), xyz.values.data_ptr(), idx.values.data_ptr()
            src.node_splits, src.edge_splits = node.row_splits.data_ptr(), idx.row_splits.data_ptr()
            src.N, src.M, src.G = nb, mb, gb
        self.desc = d
        # host node splits of the union (row counts of the members: graphs without nodes at a member's end are dropped
        # from ITS result, as its own forward would do)
        offs, cat = 0, [np.zeros(1, np.int64)]
        self.cuts = []
        g_off = 0
        for h, (nb, mb, gb) in zip(ns_host, sizes):
            cat.append(h[1:] + offs)
            rows = gb
            while rows > 0 and h[rows] == h[rows - 1]:
                rows -= 1
            self.cuts.append((g_off, g_off + rows))
            offs += nb
            g_off += gb
        # members without trailing empty graphs tile the result: their views come from one ``split`` call
        tiled = all(a[1] == b[0] for a, b in zip(self.cuts, self.cuts[1:])) and self.cuts[0][0] == 0
        self._split_sizes = ([hi - lo for lo, hi in self.cuts], self.cuts[-1][1]) if tiled else None
        known = 0
        for x in inputs_list:           # every member's index list classified by its producer, or the union is checked below
            flags = None
            for plan in x[2]._plans.values():
                if plan._flags_host is not None:
                    flags = plan._flags_host
            known = None if (known is None or flags is None) else (known | int(flags))
        if known is None:
            self.concat()               # the bind's index pass reads the union
        self.slot = FusedSchnet(route._p, depth=route.depth, gauss_args=route.gauss, fast_softplus=route.fast_softplus,
                                cfconv_flags=route.cfconv_flags, packed=route._packed)
        batch = {"z": self.z, "xyz": self.xyz, "idx": self.idx, "ns": self.ns, "es": self.es,
                 "ns_host": np.concatenate(cat)}
        self.slot.bind(batch, n, m, g, known_flags=known, arena=route._arena, work=work)
        # (not ``self.concat``: a bound method of the group inside its own slot is a reference cycle - the slot would be
        #  freed by the cycle collector some time later, not when the group is dropped, and its work set would miss the arena)
        concat, ref = _ffi.lib().mp_concat_batches, self._desc_ref
        self.slot._pre = lambda: _ffi.check(concat(ref, _ffi.stream()))
        self.slot.calls = 0
        self.edges = m

    def concat(self):
        _ffi.check(_ffi.lib().mp_concat_batches(self._desc_ref, _ffi.stream()))

    def split(self, out):
        """Member results as views of the union's (rows, 1) result."""
        full = int(out.shape[0])
        if self._split_sizes is not None and self._split_sizes[1] == full:
            return list(out.split(self._split_sizes[0]))      # one call instead of a slice per member (~2.5 us each)
        return [out[lo:min(hi, full)] for lo, hi in self.cuts]


class SchnetFusedRoute:
    """The fused forward behind ``Schnet.make_model(...)(inputs)``.

    ``tensors()`` returns the model's live weight tensors under the names of ``synth.schnet_params``; the kernels read
    them in place (the cfconv LDS images are re-packed when a weight's version counter moves).  Every distinct input
    set - identified by the storage of its five tensors and by N, M, G - owns a batch slot (``FusedSchnet``): the first
    call binds it (work buffers, one index pass + flag read) and issues the eight launches from one C-ABI call; from
    the second call on the slot's captured HIP graph is replayed.  Everything is launched on torch's current stream,
    and the returned ``(G', 1)`` tensor is a fresh copy, as a Keras model call returns a new tensor.  At most
    ``max_slots`` batches stay bound (least recently used first out); a slot keeps its input tensors alive, which is
    what makes their addresses an identity.
    """

    def __init__(self, tensors, depth, gauss_args, max_slots=8, fast_softplus=True, cfconv_flags=0):
        self._tensors = tensors
        self.depth, self.gauss = int(depth), dict(gauss_args)
        self.max_slots, self.fast_softplus, self.cfconv_flags = int(max_slots), bool(fast_softplus), int(cfconv_flags)
        self.mode = "auto"          # auto: direct launch on first sight, graph replay afterwards | graph | direct | eager
        self.copy_output = True
        self._slots = {}
        self._gslots = {}           # batch slots of the energy + force pass (fused_schnet_force.FusedSchnetForce)
        self._groups = {}           # launch groups (call_group): k bound batches served by one launch sequence
        self.max_groups = 4
        self._p = None
        self._wkey = None
        self._wlist, self._vsum, self._wcalls, self._wepoch = None, 0, 0, -1
        self._packed = None
        self._arena = WorkArena()   # recycled work-buffer sets (batches that are bound, launched once and dropped)
        self._grad_images = None    # transposed kernels / cfconv reverse images, built on the first force call
        self.single_state = True    # SchNet heads the route accepts end in one energy value per graph
        self.last = None            # how the last call ran: "direct" | "graph" | "eager"

    # -- applicability of one call -----------------------------------------------------------------------------------
    @staticmethod
    def accepts(inputs, with_forces=False):
        from .autograd import needs_grad
        from .ragged import RaggedTensor
        if not (isinstance(inputs, (list, tuple)) and len(inputs) == 3
                and all(isinstance(x, RaggedTensor) for x in inputs)):
            return False
        z, xyz, idx = (x.values for x in inputs)
        return (z.is_cuda and z.dtype in (torch.float32, torch.int64) and z.dim() == 1 and xyz.dtype == torch.float32
                and xyz.dim() == 2 and int(xyz.shape[1]) == 3 and idx.dtype == torch.int64 and idx.dim() == 2
                and int(idx.shape[1]) == 2 and z.is_contiguous() and xyz.is_contiguous() and idx.is_contiguous()
                and int(xyz.shape[0]) == int(z.shape[0]) and inputs[0].nrows() == inputs[2].nrows()
                and (with_forces or not needs_grad(z, xyz)))

    # -- weights ------------------------------------------------------------------------------------------------------
    def _sync_weights(self):
        # Fast path (every call): the version counters of the tensors seen at the last full check - in-place updates
        # (set_weights, an optimizer step) are what changes weights through this API, and they bump a counter.  The full
        # check (also notices a layer whose tensor OBJECT was replaced) walks the model: 24 us of host time per call
        # against 6 us, so it runs on every 64th call, whenever a weight tensor object was created or assigned anywhere
        # (``layers.base.weight_epoch``), and after ``release()``.  Its key holds id, storage address and version: a
        # ``tensor.data = ...`` swap (same object, same version) is seen there.
        wl = self._wlist
        if wl is not None and self._wepoch == weight_epoch():
            self._wcalls += 1
            if self._wcalls & 63:
                vs = 0
                for t in wl:
                    vs += t._version
                if vs == self._vsum:
                    return
        p = self._tensors()
        key = tuple((id(t), t.data_ptr(), t._version) for t in p.values() if t is not None)
        self._wlist = [t for t in p.values() if t is not None]
        self._vsum = sum(k[2] for k in key)
        self._wepoch = weight_epoch()
        if key == self._wkey:
            return
        moved = self._wkey is None or tuple(k[:2] for k in key) != tuple(k[:2] for k in self._wkey)
        torch.cuda.synchronize()   # forwards in flight still read the old images
        if moved:                  # other tensors: every bound slot (descriptor, graph) points at the old ones
            self._slots.clear()
            self._gslots.clear()
            self._groups.clear()
            self._arena.clear()    # cached descriptors hold the old weight addresses
            self._grad_images = None
            self._p = {k: v for k, v in p.items() if v is not None}
            self._packed = pack_weights(self._p, self.depth, int(self.gauss["bins"]))
        else:                      # same tensors, new values: re-fill the images the graphs already point at
            pack_weights(self._p, self.depth, int(self.gauss["bins"]), out=self._packed)
            if self._grad_images is not None:
                from .fused_schnet_force import make_grad_images
                make_grad_images(self._p, self.depth, int(self.gauss["bins"]), "output_mlp/0/kernel" not in self._p,
                                 out=self._grad_images)
        self._wkey = key

    # -- batch slots --------------------------------------------------------------------------------------------------
    @staticmethod
    def _key(node, xyz, idx):
        return (node.values.data_ptr(), xyz.values.data_ptr(), idx.values.data_ptr(), node.row_splits.data_ptr(),
                idx.row_splits.data_ptr(), int(node.values.shape[0]), int(idx.values.shape[0]), node.nrows(),
                idx.values._version, idx.row_splits._version, node.row_splits._version)

    def _bind(self, node, xyz, idx):
        slot = FusedSchnet(self._p, depth=self.depth, gauss_args=self.gauss, fast_softplus=self.fast_softplus,
                           cfconv_flags=self.cfconv_flags, packed=self._packed)
        batch = {"z": node.values, "xyz": xyz.values, "idx": idx.values, "ns": node.row_splits, "es": idx.row_splits,
                 "ns_host": node.row_splits_host()}
        known = None
        for plan in idx._plans.values():   # a producer (host packer, on-GPU SetRange) may have classified the list already
            if plan._flags_host is not None:
                known = plan._flags_host
        slot.bind(batch, int(node.values.shape[0]), int(idx.values.shape[0]), node.nrows(), known_flags=known,
                  arena=self._arena)
        slot.calls = 0
        return slot

    def __call__(self, inputs):
        node, xyz, idx = inputs
        self._sync_weights()
        key = self._key(node, xyz, idx)
        slot = self._slots.get(key)
        if slot is None:
            slot = self._bind(node, xyz, idx)
            while len(self._slots) >= self.max_slots:
                self._slots.pop(next(iter(self._slots)))
            self._slots[key] = slot
        elif next(reversed(self._slots)) != key:   # keep the dictionary in least-recently-used order
            self._slots[key] = self._slots.pop(key)
        slot.calls += 1
        how = self.mode
        if how == "auto":
            how = "direct" if slot.calls == 1 else "graph"
        self.last = how
        if how == "graph" and self.copy_output:
            out = slot.run_graph_fresh()   # a result buffer nobody else holds: no copy launch
            if out is not None:
                return out
        out = slot.run_current(how)
        if not self.copy_output:
            return out
        if how == "direct" and slot.calls == 1 and slot._work is not None:
            return slot.hand_out_static()   # first sight: the slot's result buffer becomes the caller's tensor, no copy
        return out.clone()

    def call_group(self, inputs_list):
        """``[model(x) for x in inputs_list]`` from ONE launch sequence (``SchnetGroup``): the batches are concatenated on the
        device and run as one union batch - the eight kernel boundaries and the weight staging are paid once for all of
        them.  First call of a group: concatenation + direct launch; later calls replay the group's HIP graph (the
        concatenation is part of it, so new coordinate values in the members' tensors are picked up).  Returns one tensor
        per member, views of a result buffer nobody else holds.  Members must be receiver-sorted batches the route
        accepts; anything else falls back to separate calls."""
        # A group that is bound already is found by its key alone (addresses, sizes, version counters of every member's
        # tensors): what ``accepts`` establishes - types, dtypes, layout - cannot change under an unchanged key, and five
        # ``accepts`` walks were half of the host time of a replayed group (46 -> 25 us per call).  Only the question whether
        # the caller wants a gradient is asked again.
        key = None
        if len(inputs_list) > 1 and self._groups:
            try:
                key = tuple(self._key(*x) for x in inputs_list)
            except (AttributeError, TypeError, ValueError):
                key = None
        grp = self._groups.get(key) if key is not None else None
        if grp is not None and torch.is_grad_enabled() and any(x[0].values.requires_grad or x[1].values.requires_grad
                                                               for x in inputs_list):
            grp = None
        if grp is None:
            inputs_list = [list(x) for x in inputs_list]
            if len(inputs_list) == 1 or not all(self.accepts(x) for x in inputs_list):
                return [self(x) for x in inputs_list]
            key = tuple(self._key(*x) for x in inputs_list)
            grp = self._groups.get(key)
        self._sync_weights()
        grp = self._groups.get(key)        # (a weight change that moved tensors has cleared the groups)
        if grp is None:
            grp = SchnetGroup(self, inputs_list)
            while len(self._groups) >= self.max_groups:
                self._groups.pop(next(iter(self._groups)))
            self._groups[key] = grp
        elif next(reversed(self._groups)) != key:
            self._groups[key] = self._groups.pop(key)
        slot = grp.slot
        if not slot.sorted:              # the union of unsorted members: the separate route handles those
            del self._groups[key]
            return [self(x) for x in inputs_list]
        slot.calls += 1
        how = self.mode
        if how == "auto":
            how = "direct" if slot.calls == 1 else "graph"
        self.last = how
        if how == "graph":
            out = slot.run_graph_fresh()
            if out is not None:
                return grp.split(out)
        out = slot.run_current(how)
        if how == "direct" and slot.calls == 1 and slot._work is not None:
            return grp.split(slot.hand_out_static())   # first sight: the slot's result buffer becomes the callers' tensor
        return grp.split(out.clone())

    def energy_force(self, inputs):
        """``(energy (G', 1), force (N, 3))`` with force = -dE/dx: fused forward + hand-written reverse pass, one HIP
        graph per bound batch (gcnn_keras_amd/fused_schnet_force.py)."""
        from .fused_schnet_force import FusedSchnetForce, make_grad_images
        node, xyz, idx = inputs
        self._sync_weights()
        if self._grad_images is None:
            self._grad_images = make_grad_images(self._p, self.depth, int(self.gauss["bins"]),
                                                 "output_mlp/0/kernel" not in self._p)
        key = self._key(node, xyz, idx)
        slot = self._gslots.get(key)
        if slot is None:
            slot = FusedSchnetForce(self._p, self._packed, self._grad_images, self.depth, self.gauss,
                                    fast_softplus=self.fast_softplus, cfconv_flags=self.cfconv_flags)
            slot.bind(node, xyz, idx)
            while len(self._gslots) >= self.max_slots:
                self._gslots.pop(next(iter(self._gslots)))
            self._gslots[key] = slot
        elif next(reversed(self._gslots)) != key:
            self._gslots[key] = self._gslots.pop(key)
        slot.calls += 1
        how = self.mode
        if how == "auto":
            how = "eager" if slot.calls == 1 else "graph"
        elif how == "direct":
            how = "eager"
        self.last = how
        if how == "graph" and self.copy_output:
            got = slot.run_graph_fresh()   # an (energy, force) pair nobody else holds: no copy launches
            if got is not None:
                return got
        eng, force = slot.run_current(how)
        return (eng.clone(), force.clone()) if self.copy_output else (eng, force)

    def slot_of(self, inputs):
        return self._slots.get(self._key(*inputs))

    def check_flags(self):
        for slot in list(self._slots.values()) + list(self._gslots.values()) + [g.slot for g in self._groups.values()]:
            slot.check_flags()

    def release(self):
        """Unbind every batch (frees the work buffers and the references to the input tensors)."""
        torch.cuda.synchronize()
        self._slots.clear()
        self._gslots.clear()
        self._groups.clear()
        self._arena.clear()
        self._wlist = None
