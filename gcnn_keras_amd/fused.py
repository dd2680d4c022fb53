"""Fused SchNet forward: the arithmetic of kgcnn/literature/Schnet.py:104-148 in eight kernels, replayed from a HIP graph.

    stage0              edge_prepare (index shift, receiver/sender split, flags, distance) and node_in
                        (Embedding -> Dense(64->128) -> Dense_nobias) on disjoint workgroups             (1 launch)
    per block:          cfconv_gauss_fused (Gauss basis, filter MLP, gather, multiply, segment-sum)     (depth launches)
                        node_update / node_last (2-3 chained Dense on the node tile, residual)         (depth launches)
    readout             PoolingNodes(sum) + output MLP                                                 (1 launch)

Every buffer is allocated when a batch is bound; ``forward()`` only replays the captured graph.
"""
import ctypes

import numpy as np
import torch

from . import _ffi


def supports(config):
    """True if a Schnet.make_model configuration maps onto the fused kernels (else use the layer path)."""
    ia = config["interaction_args"]
    return (ia.get("units") == 128 and ia.get("cfconv_pool") in ("sum", "segment_sum", "reduce_sum")
            and ia.get("activation") in ("kgcnn>shifted_softplus", "shifted_softplus") and ia.get("use_bias", True)
            and config["input_embedding"]["node"]["output_dim"] == 64
            and list(config["last_mlp"]["units"]) == [128, 64]
            and list(config["output_mlp"]["units"]) == [64, 1]
            and config["node_pooling_args"].get("pooling_method") in ("sum", "segment_sum", "reduce_sum")
            and config.get("output_embedding", "graph") == "graph"
            and int(config["gauss_args"]["bins"]) <= 32)


class FusedSchnet:
    def __init__(self, params, depth=3, gauss_args=None, fast_softplus=True, use_graph=True, cfconv_flags=0):
        if not torch.cuda.is_available():
            raise _ffi.EngineError("FusedSchnet needs an MI355X (no CPU fallback)")
        self.depth = depth
        self.gauss = dict(gauss_args or {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4})
        self.flags_arg = (1 if fast_softplus else 0) | int(cfconv_flags)
        self._stream_ptr = None
        self._desc = None
        self.use_graph = use_graph
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in params.items()}
        if tuple(self.p["embedding"].shape)[1] != 64 or tuple(self.p["dense0/kernel"].shape) != (64, 128):
            raise ValueError("FusedSchnet is built for embedding width 64 and 128 units")
        # filter-MLP weights of every block in the cfconv kernel's LDS image order (packed once per weight update)
        nfl = _ffi.lib().mp_cfconv_packed_floats()
        self.packed = []
        for i in range(depth):
            pre = "interaction%d/cfconv/" % i
            buf = torch.empty(nfl, dtype=torch.float32, device="cuda")
            _ffi.call("mp_cfconv_pack_f32", _ffi.ptr(self.p[pre + "dense1/kernel"]),
                      _ffi.ptr(self.p.get(pre + "dense1/bias")), int(self.gauss["bins"]),
                      _ffi.ptr(self.p[pre + "dense2/kernel"]), _ffi.ptr(self.p.get(pre + "dense2/bias")),
                      _ffi.ptr(buf), _ffi.stream())
            self.packed.append(buf)
        torch.cuda.synchronize()
        self.stream = torch.cuda.Stream()
        self.graph = None
        self.num_launches = 1 + 2 * depth + 1
        self._b = None

    # ------------------------------------------------------------------------------------------------ binding
    def bind(self, b, n, m, g):
        """Attach a resident batch (dict of device tensors: z, xyz, idx, ns, es) and allocate all work buffers."""
        self._b, self.N, self.M, self.G = b, n, m, g
        dev = "cuda"
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
        self._prepare()
        torch.cuda.synchronize()
        f = int(self.flags.item())
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
        self.flags.zero_()
        splits = np.asarray(b["ns_host"])
        rows = g
        while rows > 0 and splits[rows] == splits[rows - 1]:
            rows -= 1
        self.out_rows = rows  # tf.math.segment_sum drops trailing empty graphs (kgcnn/layers/pooling.py:215-219)
        self.graph = None
        self._desc = None
        torch.cuda.synchronize()

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

    def _launch_all(self):
        # One linear chain on one stream.  (A two-branch graph - node_in beside edge_prepare - was measured 9 % SLOWER
        # at config 2: the fork/join costs more than the ~5 us of overlap it buys.)
        p, b = self.p, self._b
        if self.sorted or self.M == 0:
            # stage 0: node-input chain and edge preparation in one launch (independent work on disjoint workgroups)
            _ffi.call("mp_schnet_stage0_f32", _ffi.ptr(b["z"]), self.N, _ffi.ptr(p["embedding"]),
                      int(p["embedding"].shape[0]), 64, _ffi.ptr(p["dense0/kernel"]), _ffi.ptr(p.get("dense0/bias")),
                      _ffi.ptr(p["interaction0/dense1/kernel"]), _ffi.ptr(self.n), _ffi.ptr(self.x),
                      _ffi.ptr(b["idx"]), self.M, _ffi.ptr(b["ns"]), _ffi.ptr(b["es"]), self.G, _ffi.ptr(b["xyz"]),
                      _ffi.ptr(self.recv), _ffi.ptr(self.send), _ffi.ptr(self.dist), _ffi.ptr(self.flags),
                      self.flags_arg & 1, _ffi.stream())
        else:
            self._prepare()
            _ffi.call("mp_sort_segments_i32", _ffi.ptr(self.recv), self.M, _ffi.ptr(self.recv_sorted),
                      _ffi.ptr(self.perm), _ffi.ptr(self.sort_ws), self.sort_ws_bytes, _ffi.stream())
            _ffi.call("mp_schnet_node_in_f32", _ffi.ptr(b["z"]), self.N, _ffi.ptr(p["embedding"]),
                      int(p["embedding"].shape[0]), 64, _ffi.ptr(p["dense0/kernel"]), _ffi.ptr(p.get("dense0/bias")),
                      _ffi.ptr(p["interaction0/dense1/kernel"]), _ffi.ptr(self.n), _ffi.ptr(self.x),
                      self.flags_arg & 1, _ffi.stream())
        for i in range(self.depth):
            pre = "interaction%d/" % i
            self._cfconv(i, self.agg)
            if i + 1 < self.depth:
                _ffi.call("mp_schnet_node_update_f32", _ffi.ptr(self.agg), self.N, _ffi.ptr(p[pre + "dense2/kernel"]),
                          _ffi.ptr(p.get(pre + "dense2/bias")), _ffi.ptr(p[pre + "dense3/kernel"]),
                          _ffi.ptr(p.get(pre + "dense3/bias")), _ffi.ptr(self.n),
                          _ffi.ptr(p["interaction%d/dense1/kernel" % (i + 1)]), _ffi.ptr(self.x), self.flags_arg & 1,
                          _ffi.stream())
            else:
                _ffi.call("mp_schnet_node_last_f32", _ffi.ptr(self.agg), self.N, _ffi.ptr(p[pre + "dense2/kernel"]),
                          _ffi.ptr(p.get(pre + "dense2/bias")), _ffi.ptr(p[pre + "dense3/kernel"]),
                          _ffi.ptr(p.get(pre + "dense3/bias")), _ffi.ptr(self.n), _ffi.ptr(p["last_mlp/0/kernel"]),
                          _ffi.ptr(p.get("last_mlp/0/bias")), _ffi.ptr(p["last_mlp/1/kernel"]),
                          _ffi.ptr(p.get("last_mlp/1/bias")), _ffi.ptr(self.h), self.flags_arg & 1, _ffi.stream())
        _ffi.call("mp_schnet_readout_f32", _ffi.ptr(self.h), _ffi.ptr(b["ns"]), self.G,
                  _ffi.ptr(p["output_mlp/0/kernel"]), _ffi.ptr(p.get("output_mlp/0/bias")),
                  _ffi.ptr(p["output_mlp/1/kernel"]), _ffi.ptr(p.get("output_mlp/1/bias")), _ffi.ptr(self.out),
                  _ffi.stream())

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
        p, b, ga = self.p, self._b, self.gauss
        d = _ffi.SchnetForwardDesc()
        d.N, d.M, d.G = self.N, self.M, self.G
        d.depth, d.vocab, d.flags, d.bins = self.depth, int(p["embedding"].shape[0]), self.flags_arg, int(ga["bins"])
        d.g_distance, d.g_sigma, d.g_offset = float(ga["distance"]), float(ga["sigma"]), float(ga["offset"])
        addr = lambda t: None if t is None else t.data_ptr()
        d.numbers, d.xyz, d.idx = addr(b["z"]), addr(b["xyz"]), addr(b["idx"])
        d.node_splits, d.edge_splits = addr(b["ns"]), addr(b["es"])
        d.embedding, d.W0, d.b0 = addr(p["embedding"]), addr(p["dense0/kernel"]), addr(p.get("dense0/bias"))
        for i in range(self.depth):
            pre = "interaction%d/" % i
            d.Wx[i], d.packed[i] = addr(p[pre + "dense1/kernel"]), addr(self.packed[i])
            d.W2[i], d.b2[i] = addr(p[pre + "dense2/kernel"]), addr(p.get(pre + "dense2/bias"))
            d.W3[i], d.b3[i] = addr(p[pre + "dense3/kernel"]), addr(p.get(pre + "dense3/bias"))
        d.Wl0, d.bl0 = addr(p["last_mlp/0/kernel"]), addr(p.get("last_mlp/0/bias"))
        d.Wl1, d.bl1 = addr(p["last_mlp/1/kernel"]), addr(p.get("last_mlp/1/bias"))
        d.Wo0, d.bo0 = addr(p["output_mlp/0/kernel"]), addr(p.get("output_mlp/0/bias"))
        d.Wo1, d.bo1 = addr(p["output_mlp/1/kernel"]), addr(p.get("output_mlp/1/bias"))
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
        if self._desc is None:
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
        with torch.cuda.stream(self.stream):
            ms = _HipTimer().time_ms(lambda: self._cfconv(0, scratch), iters)
        torch.cuda.synchronize()
        flops = float(self.M) * (2.0 * (bins * 128 + 128 * 128) + 2.0 * 128)
        # algorithmic bytes of one launch: distance + ids per edge, sender rows once, output rows once
        alg_bytes = float(self.M) * (4 + 8) + 2.0 * self.N * 128 * 4
        achieved = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "cfconv_fused_kernel<4,gauss> (SchNetCFconv, one interaction block)",
                "achieved": achieved, "peak": mfma_peak_tf, "unit": "TFLOP/s", "frac": achieved / mfma_peak_tf,
                "traffic": None, "avg_launch_us": ms * 1e3, "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": alg_bytes,
                "hbm_gbs_at_this_rate": alg_bytes / (ms * 1e-3) / 1e9, "hbm_peak_gbs": hbm_peak_gbs}

    def __del__(self):
        try:
            if self.graph is not None:
                _ffi.call("mp_graph_destroy", self.graph)
        except Exception:
            pass
