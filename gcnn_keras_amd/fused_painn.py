"""Fused PaiNN forward and energy + force pass (kgcnn/literature/PAiNN.py:100-155 inside kgcnn/model/force.py:159-201).

Per block three FP32-MFMA launches (``mp_dense_chain_f32``, csrc/mp_chain.hip: the two Dense layers of the filter net
and of the update net each run as ONE launch with the weights in registers and the 128-wide intermediate in LDS;
activations, their derivatives and residual adds ride in the epilogues) and three memory-bound kernels
(csrc/mp_painn_fused.hip): ~22 launches for a depth-3 forward and ~20 more for the reverse pass that yields dE/dx, all
replayed from ONE HIP graph per bound batch.  The reverse pass is written out kernel by kernel here - no tape: every
saved tensor is a buffer of the batch slot.

    stage0    Embedding, EquivariantInitialize, index pass, r_ij, d, Bessel basis (+ d rbf / d d when forces are wanted)
    block i   h1 = z W1 + b1 | s = act(h1) Wphi + bphi | message kernel (z' = z + ds, v' = v + dv)
              uv = v' [Wu|Wv] | c = [z'|norm], prod | h2 = c Wd + bd | a = act(h2) Wa + ba | z'' , v''
    readout   PoolingNodes(sum) -> output MLP
    reverse   readout^T | per block, last to first: post^T, Wa^T, Wd^T (x act'), pre^T, [Wu|Wv]^T, message^T
              (sender-parallel; per-edge dE/dd, dE/dr_ij), Wphi^T, W1^T (x act') | geometry^T -> forces
"""
import ctypes
import os

import numpy as np
import torch

from . import _ffi
from .layers.base import weight_epoch
from .result_ring import ResultRing

_SUM = ("sum", "segment_sum", "reduce_sum")
_CONST_INIT = {"zeros": 0.0, "eps": 1e-7, "ones": 1.0}


def _as_list(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def supports(config):
    """True if a ``PAiNN.make_model`` configuration (merged keyword dictionary) maps onto the fused pipeline."""
    try:
        inputs, conv, upd, om = config["inputs"], config["conv_args"], config["update_args"], config["output_mlp"]
        if len(inputs) != 3 or len(inputs[0]["shape"]) != 1 or tuple(inputs[1]["shape"])[-1] != 3:
            return False
        units = _as_list(om["units"], 1)
        acts = _as_list(om.get("activation"), len(units))
        init = config["equiv_initialize_kwargs"]
        for a in [conv.get("activation", "swish"), upd.get("activation", "swish")] + acts:
            _ffi.activation_code(a)
        return bool(
            conv.get("units") == 128 and upd.get("units") == 128 and conv.get("conv_pool", "sum") in _SUM
            and conv.get("use_bias", True) is True and upd.get("use_bias", True) is True
            and config["input_embedding"]["node"]["output_dim"] == 128
            and not config.get("equiv_normalization") and not config.get("node_normalization")
            and config.get("output_embedding", "graph") == "graph"
            and config["pooling_args"].get("pooling_method") in _SUM
            and int(init.get("dim", 3)) == 3 and (init.get("method", "zeros") in _CONST_INIT or init.get("method") == "const")
            and 1 <= int(config["bessel_basis"]["num_radial"]) <= 32
            and int(config["bessel_basis"]["envelope_exponent"]) >= 1
            and acts[-1] in ("linear", None) and all(_as_list(om.get("use_bias", True), len(units)))
            and int(config["depth"]) >= 1)
    except (KeyError, TypeError, IndexError, ValueError):
        return False


def make_images(p, depth, n_out, out=None):
    """Weight layouts the reverse pass and the fused update need, made once per weight update: ``[Wu | Wv]`` side by side
    (one GEMM yields v_u and v_v) and the transposed kernels.  ``out`` re-fills existing images in place."""
    def put(name, value):
        if out is not None:
            out[name].copy_(value)
        else:
            images[name] = value.contiguous().clone()

    keep = []

    def pack(name, value):
        """``mp_chain_pack_f32`` image (register-slice order of csrc/mp_chain.hip) of a (K, U) matrix."""
        value = value.contiguous()
        keep.append(value)
        if out is None:
            images[name] = torch.empty(value.numel(), dtype=torch.float32, device=value.device)
        _ffi.call("mp_chain_pack_f32", _ffi.ptr(value), int(value.shape[0]), int(value.shape[1]), _ffi.ptr(images[name]),
                  _ffi.stream())

    images = {} if out is None else out
    for i in range(depth):
        c, u = "conv%d/" % i, "update%d/" % i
        uv = torch.cat([p[u + "lin_u/kernel"], p[u + "lin_v/kernel"]], dim=1)
        pack("uv%d/P" % i, uv)
        pack("uvT%d/P" % i, uv.t())
        for name in (c + "dense1", c + "phi", u + "dense1", u + "a"):
            pack(name + "/P", p[name + "/kernel"])
            pack(name + "/TP", p[name + "/kernel"].t())
        # bf16-piece operand image of the filter Dense (w = Dense(3F)(rbf), painn_conv.py:100) for the matrix-pipe message
        # kernels; needs a free k slot for the bias (basis size <= 31)
        wk = p[c + "w/kernel"]
        if int(wk.shape[0]) <= 31:
            if out is None:
                images[c + "w/F"] = torch.empty(_ffi.MP_PAINN_FILTER_IMAGE_BYTES, dtype=torch.uint8, device=wk.device)
            _ffi.call("mp_painn_filter_pack_f32", _ffi.ptr(wk), _ffi.ptr(p.get(c + "w/bias")), int(wk.shape[0]),
                      _ffi.ptr(images[c + "w/F"]), _ffi.stream())
    for k in range(n_out):
        put("output_mlp/%d/T" % k, p["output_mlp/%d/kernel" % k].t())
    torch.cuda.current_stream().synchronize()   # the packed sources in `keep` are temporaries
    return images


def message_tile_table(node_splits, ptr, basis, per=None, lds_limit=160 * 1024, reverse=False):
    """Tile table of ``mp_painn_message_tiles_f32`` (csrc/mp_painn_fused.hip) from host arrays: ``(T, 8)`` int32 rows
    ``{r_lo, r_hi, s_lo, s_hi, e_lo, e_hi, 0, 0}`` - a few consecutive nodes ``[r_lo, r_hi)`` of ONE graph, the graph's node
    range ``[s_lo, s_hi)`` (whose s / v rows the workgroup stages: a tile's senders lie in its own graph) and the tile's
    edge range ``[ptr[r_lo], ptr[r_hi])`` in the receiver CSR.  ``per`` nodes per tile (default: enough tiles for two
    workgroups per CU, at least 2, at most 62).  Returns ``{"table", "count", "max_rows", "max_edges"}`` or None when a
    graph's rows and a tile's edge data do not fit ``lds_limit`` bytes of LDS.  ``reverse``: the table of
    ``mp_painn_message_bwd_tiles_f32`` - the same rows over SENDERS and the sender CSR ``ptr1`` (LDS need of that kernel)."""
    ns = np.asarray(node_splits, dtype=np.int64)
    ptr = np.asarray(ptr, dtype=np.int64)
    n = int(ns[-1]) if len(ns) else 0
    if n == 0 or basis > 31:
        return None
    sizes = ns[1:] - ns[:-1]
    # forward: two receivers per wave step, tiles for two workgroups per CU; reverse: every sender is a serial chain of
    # its wave (~3 us), so as few senders per tile as one round of 512 resident workgroups allows
    # (two or three: three senders with their graph's gradient rows are ~77 KB of LDS, two workgroups per CU; larger
    # batches take more rounds of the same tiles)
    per = int(per) if per else (min(3, max(2, int(-(-n // 448)))) if reverse else max(2, 2 * int(-(-n // 1024))))
    per = max(1, min(per, 62))
    count = -(-sizes // per)                                      # tiles per graph
    total = int(count.sum())
    if total == 0:
        return None
    graph = np.repeat(np.arange(len(sizes)), count)
    first = np.cumsum(count) - count
    k = np.arange(total) - np.repeat(first, count)
    r_lo = ns[graph] + k * per
    r_hi = np.minimum(r_lo + per, ns[graph + 1])
    table = np.zeros((total, 8), dtype=np.int32)
    table[:, 0], table[:, 1], table[:, 2], table[:, 3] = r_lo, r_hi, ns[graph], ns[graph + 1]
    table[:, 4], table[:, 5] = ptr[r_lo], ptr[r_hi]
    max_rows, max_edges = int(sizes.max()), int((table[:, 5] - table[:, 4]).max())
    lds = ctypes.c_size_t(0)
    max_own = int((r_hi - r_lo).max())
    if reverse:
        _ffi.call("mp_painn_message_bwd_tiles_lds_bytes", max_rows, max_own, max_edges, int(basis), 1, ctypes.byref(lds))
    else:
        _ffi.call("mp_painn_message_tiles_lds_bytes", max_rows, max_edges, int(basis), 1, ctypes.byref(lds))
    if lds.value > lds_limit:
        return None
    return {"table": table, "count": total, "max_rows": max_rows, "max_edges": max_edges, "max_own": max_own}


class FusedPainn:
    """One batch slot: buffers, the two captured graphs (energy only; energy + forces) of ONE bound batch."""

    def __init__(self, p, images, cfg):
        if not torch.cuda.is_available():
            raise _ffi.EngineError("FusedPainn needs an MI355X (no CPU fallback)")
        self.p, self.w, self.cfg = p, images, cfg
        self.depth = int(cfg["depth"])
        self.B = int(cfg["bessel_basis"]["num_radial"])
        self.act_conv = _ffi.activation_code(cfg["conv_args"].get("activation", "swish"))
        self.act_upd = _ffi.activation_code(cfg["update_args"].get("activation", "swish"))
        units = _as_list(cfg["output_mlp"]["units"], 1)
        self.out_units = [int(u) for u in units]
        self.act_out = [_ffi.activation_code(a) for a in _as_list(cfg["output_mlp"].get("activation"), len(units))]
        init = cfg["equiv_initialize_kwargs"]
        self.v_init = float(init.get("value", 1.0)) if init.get("method") == "const" else _CONST_INIT[init.get("method", "zeros")]
        cutoff = cfg["conv_args"].get("cutoff")
        self.cos_cutoff = float(cutoff) if cutoff is not None else -1.0
        # readout in one launch (mp_pool_mlp2_f32) for the two-layer heads [H, 1]; other heads: pool + Dense launches
        self.fast_readout = len(self.out_units) == 2 and self.out_units[0] in (64, 128) and self.out_units[1] == 1
        self.stream = torch.cuda.Stream()
        self.graphs = {}
        self.rings = {}
        self.calls = 0
        self._pre = None      # launch group: the concatenation of the member batches runs in front of every pass

    # ------------------------------------------------------------------------------------------------ binding
    def bind(self, node, xyz, idx, grad):
        """Attach resident ragged inputs; allocates every buffer of the forward (and of the reverse pass if ``grad``)."""
        self.inputs = (node, xyz, idx)
        self.grad = bool(grad)
        n, m, g = int(node.values.shape[0]), int(idx.values.shape[0]), node.nrows()
        self.N, self.M, self.G = n, m, g
        dev, f32 = node.values.device, torch.float32
        e = lambda *shape: torch.empty(shape, dtype=f32, device=dev)
        plan = idx.index_plan(node)          # CSRs of both columns (receiver: forward; sender: reverse pass)
        if plan.flags_host() & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")
        self.ptr0, self.perm0, _ = plan.csr(0)
        self.ptr1, self.perm1, _ = plan.csr(1) if self.grad else (None, None, None)
        self.tiles0 = self._tile_table(node, self.ptr0, self.perm0)
        # the reverse kernel gathers its tile's edge data through perm1: any edge order
        self.tiles1 = self._tile_table(node, self.ptr1, None, reverse=True) if self.grad else None
        if os.environ.get("MPENGINE_PAINN_BWD_TILES") == "0":
            self.tiles1 = None
        mm = max(m, 1)
        self.recv = torch.empty(mm, dtype=torch.int32, device=dev)
        self.send = torch.empty(mm, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self.dist, self.rij, self.rbf = e(mm), e(mm, 3), e(mm, self.B)
        self.rbfd = e(mm, self.B) if self.grad else None
        self.env = e(mm) if self.cos_cutoff > 0 else None
        self.envd = e(mm) if (self.cos_cutoff > 0 and self.grad) else None
        self.z0, self.v0 = e(n, 128), e(n, 3, 128)
        nblk = self.depth if self.grad else 1   # the reverse pass needs every block's intermediates
        self.blk = [{"h1": e(n, 128), "s": e(n, 384), "zp": e(n, 128), "vp": e(n, 3, 128),
                     "uv": e(3 * n, 256), "c": e(n, 256), "prod": e(n, 128), "h2": e(n, 128),
                     "a": e(n, 384)} for _ in range(nblk)]
        nzv = self.depth if self.grad else 2    # block outputs: all of them (v_in of the next block is saved) or ping-pong
        self.zs = [e(n, 128) for _ in range(nzv)]
        self.vs = [e(n, 3, 128) for _ in range(nzv)]
        self.pooled = e(g, 128)
        self.pre_out = [e(g, u) for u in self.out_units]
        self.post_out = [e(g, u) for u in self.out_units]
        splits = node.row_splits_host()
        rows = g
        while rows > 0 and splits[rows] == splits[rows - 1]:
            rows -= 1
        self.out_rows = rows   # tf.math.segment_sum drops trailing empty graphs (kgcnn/layers/pooling.py:215-219)
        if self.grad:
            if self.out_units[-1] != 1:
                raise ValueError("fused forces are built for one energy state")
            self.ones = torch.ones((g, 1), dtype=f32, device=dev)
            self.g_out = [e(g, u) for u in ([128] + self.out_units[:-1])]   # dE/d(input of output layer k)
            self.gz, self.gv = e(n, 128), e(n, 3, 128)
            self.g_a, self.g_prod, self.g_c = e(n, 384), e(n, 128), e(n, 256)
            self.g_zp, self.g_uv, self.g_vp = e(n, 128), e(3 * n, 256), e(n, 3, 128)
            self.g_s = e(n, 384)
            self.g_d, self.g_rij = e(2, mm), e(2, mm, 3)   # one slice per feature half of the reverse message kernel
            self.force = e(n, 3)
        self.graphs = {}
        self.rings = {}

    def _tile_table(self, node, ptr, perm, reverse=False):
        """Tiles of the LDS-staged message kernel for this batch (``message_tile_table``), on the device; None when the
        kernel does not apply: permuted (unsorted) edge lists, basis sizes without a free bias slot, graphs whose node
        rows do not fit LDS, ``MPENGINE_PAINN_TILES=0``."""
        n = int(node.values.shape[0])
        if perm is not None or self.B > 31 or n == 0 or self.M == 0 or os.environ.get("MPENGINE_PAINN_TILES") == "0":
            return None
        # the CSR is read back once per bound batch: a few KB, next to the flag word bind reads anyway
        tl = message_tile_table(node.row_splits_host(), ptr.cpu().numpy(), self.B,
                                per=int(os.environ.get("MPENGINE_PAINN_TILE_R", "0")) or None, reverse=reverse)
        if tl is None:
            return None
        tl["table"] = torch.from_numpy(tl["table"]).to(node.values.device)
        return tl

    # ------------------------------------------------------------------------------------------------ launches
    @staticmethod
    def _dense(x, rows, k, w, b, u, out, act=0, out_pre=None, grad_act=0, grad_pre=None, addend=None):
        """out = act(x W + b) [* grad_act'(grad_pre)] [+ addend]; out_pre additionally keeps x W + b."""
        _ffi.call("mp_dense_ex_f32", _ffi.ptr(x), rows, k, _ffi.ptr(w), _ffi.ptr(b), u, act, 0.0, 0, grad_act, 0.0,
                  None, _ffi.ptr(addend), _ffi.ptr(out_pre), _ffi.ptr(grad_pre), _ffi.ptr(out), _ffi.stream())

    @staticmethod
    def _chain(x, rows, k1, w1, b1, u1, out, act=0, save_pre=None, grad_pre=None, w2=None, b2=None, u2=0, addend=None):
        """One or two Dense layers in one launch (csrc/mp_chain.hip): m = act(x W1 + b1) [save_pre keeps x W1 + b1], or
        m = (x W1) * act'(grad_pre) in the reverse pass; out = m W2 + b2 + addend (W2 absent: out = m + addend)."""
        _ffi.call("mp_dense_chain_f32", _ffi.ptr(x), rows, k1, _ffi.ptr(w1), _ffi.ptr(b1), u1, act, 0.0,
                  _ffi.ptr(save_pre), _ffi.ptr(grad_pre), _ffi.ptr(w2), _ffi.ptr(b2), u2, _ffi.ptr(addend), _ffi.ptr(out),
                  _ffi.stream())

    def _forward(self, with_forces=False):
        p, w, n, m = self.p, self.w, self.N, self.M
        node, xyz, idx = self.inputs
        _ffi.call("mp_painn_stage0_f32", _ffi.ptr(node.values), 1 if node.values.dtype == torch.int64 else 0, n,
                  _ffi.ptr(p["embedding"]),
                  int(p["embedding"].shape[0]), self.v_init, _ffi.ptr(self.z0), _ffi.ptr(self.v0), _ffi.ptr(idx.values), m,
                  _ffi.ptr(node.row_splits), _ffi.ptr(idx.row_splits), self.G, _ffi.ptr(xyz.values),
                  _ffi.ptr(p["bessel/frequencies"]), self.B, float(self.cfg["bessel_basis"]["cutoff"]),
                  int(self.cfg["bessel_basis"]["envelope_exponent"]), self.cos_cutoff, _ffi.ptr(self.recv),
                  _ffi.ptr(self.send), _ffi.ptr(self.flags), _ffi.ptr(self.dist), _ffi.ptr(self.rij), _ffi.ptr(self.rbf),
                  _ffi.ptr(self.rbfd), _ffi.ptr(self.env), _ffi.ptr(self.envd), _ffi.stream())
        z, v = self.z0, self.v0
        for i in range(self.depth):
            c, u = "conv%d/" % i, "update%d/" % i
            t = self.blk[i if self.grad else 0]
            z_out, v_out = self.zs[i % len(self.zs)], self.vs[i % len(self.vs)]
            # Dense(act) writes the activation and (for the reverse pass) keeps the pre-activation h1 beside it
            self._chain(z, n, 128, w[c + "dense1/P"], p.get(c + "dense1/bias"), 128, t["s"], act=self.act_conv,
                        save_pre=t["h1"] if self.grad else None, w2=w[c + "phi/P"], b2=p.get(c + "phi/bias"), u2=384)
            tl = self.tiles0
            if tl is not None and (c + "w/F") in w:   # node tiles in LDS, filter on the matrix pipe
                _ffi.call("mp_painn_message_tiles_f32", _ffi.ptr(t["s"]), _ffi.ptr(v), n, _ffi.ptr(self.rbf), self.B,
                          _ffi.ptr(self.env), _ffi.ptr(self.rij), _ffi.ptr(w[c + "w/F"]), _ffi.ptr(self.ptr0),
                          _ffi.ptr(self.send), m, _ffi.ptr(tl["table"]), tl["count"], tl["max_rows"], tl["max_edges"],
                          _ffi.ptr(z), _ffi.ptr(t["zp"]), _ffi.ptr(t["vp"]), _ffi.stream())
            else:
                _ffi.call("mp_painn_message_f32", _ffi.ptr(t["s"]), _ffi.ptr(v), n, _ffi.ptr(self.rbf), self.B,
                          _ffi.ptr(self.env), _ffi.ptr(self.rij), _ffi.ptr(p[c + "w/kernel"]),
                          _ffi.ptr(p.get(c + "w/bias")), _ffi.ptr(self.ptr0), _ffi.ptr(self.perm0), _ffi.ptr(self.send), m,
                          _ffi.ptr(z), _ffi.ptr(t["zp"]), _ffi.ptr(t["vp"]), _ffi.stream())
            self._chain(t["vp"], 3 * n, 128, w["uv%d/P" % i], None, 256, t["uv"])
            # PAiNNUpdate's element-wise steps (norm / scalar product before, the gated residual updates after) are the
            # prologue and the epilogue of its two-layer chain: one launch; c, prod, a reach HBM for the reverse pass only
            g = self.grad
            _ffi.call("mp_painn_update_fused_f32", _ffi.ptr(t["zp"]), _ffi.ptr(t["vp"]), _ffi.ptr(t["uv"]), n,
                      _ffi.ptr(w[u + "dense1/P"]), _ffi.ptr(p.get(u + "dense1/bias")), self.act_upd, 0.0,
                      _ffi.ptr(t["h2"]) if g else None, _ffi.ptr(w[u + "a/P"]), _ffi.ptr(p.get(u + "a/bias")),
                      _ffi.ptr(t["c"]) if g else None, _ffi.ptr(t["prod"]) if g else None,
                      _ffi.ptr(t["a"]) if g else None, _ffi.ptr(z_out), _ffi.ptr(v_out), _ffi.stream())
            z, v = z_out, v_out
        if self.fast_readout:   # PoolingNodes(sum) + MLP([H, 1]) - and, for forces, dE/dz of the readout - in one launch
            _ffi.call("mp_pool_mlp2_f32", _ffi.ptr(z), _ffi.ptr(node.row_splits), self.G, 128,
                      _ffi.ptr(p["output_mlp/0/kernel"]), _ffi.ptr(p.get("output_mlp/0/bias")), self.out_units[0],
                      self.act_out[0], 0.0, _ffi.ptr(p["output_mlp/1/kernel"]), _ffi.ptr(p.get("output_mlp/1/bias")),
                      _ffi.ptr(self.post_out[-1]), _ffi.ptr(self.gz) if with_forces else None, _ffi.stream())
            return
        # PoolingNodes(sum) + output MLP (PAiNN.py:146-147); every layer keeps its pre-activation
        _ffi.call("mp_pool_graph_f32", _ffi.MP_SUM, _ffi.ptr(z), _ffi.ptr(node.row_splits), self.G, 128, None,
                  _ffi.ptr(self.pooled), _ffi.stream())
        x, k_in = self.pooled, 128
        last = len(self.out_units) - 1
        for k, units in enumerate(self.out_units):   # act_out[last] is linear: post_out[last] is the energy
            self._dense(x, self.G, k_in, p["output_mlp/%d/kernel" % k], p.get("output_mlp/%d/bias" % k), units,
                        self.post_out[k], act=self.act_out[k],
                        out_pre=self.pre_out[k] if (self.grad and k < last) else None)
            x, k_in = self.post_out[k], units

    def _backward(self):
        """dE/dx for one energy state: the forward's kernels mirrored, last to first (kgcnn/model/force.py:159-177 computes
        the same derivative with a GradientTape)."""
        p, w, n, m = self.p, self.w, self.N, self.M
        node = self.inputs[0]
        last = len(self.out_units) - 1
        t, k_in = self.ones, self.out_units[-1]
        for k in range(last, -1, -1) if not self.fast_readout else ():   # fast readout: the forward wrote gz already
            # t = dE/d pre_k -> dE/d pre_{k-1} = (t W_k^T) * act_{k-1}'(pre_{k-1})
            width = 128 if k == 0 else self.out_units[k - 1]
            self._dense(t, self.G, k_in, w["output_mlp/%d/T" % k], None, width, self.g_out[k],
                        grad_act=self.act_out[k - 1] if k > 0 else 0, grad_pre=self.pre_out[k - 1] if k > 0 else None)
            t, k_in = self.g_out[k], width
        if not self.fast_readout:
            _ffi.call("mp_repeat_rows_f32", _ffi.ptr(t), _ffi.ptr(node.row_splits), self.G, 128, n, _ffi.ptr(self.gz),
                      _ffi.stream())
        tiles = self.tiles1 is not None and all(("conv%d/w/F" % i) in w for i in range(self.depth))
        # the readout sees z only: no gradient reaches the last block's v'' (null pointer = zeros, no fill launch)
        for i in range(self.depth - 1, -1, -1):
            c, u = "conv%d/" % i, "update%d/" % i
            b = self.blk[i]
            v_in = self.v0 if i == 0 else self.vs[i - 1]
            # reverse of the fused PAiNNUpdate: post_bwd / pre_bwd as prologue / epilogue of the transposed chain
            gv_in = self.gv if i < self.depth - 1 else None
            _ffi.call("mp_painn_update_fused_bwd_f32", _ffi.ptr(self.gz), _ffi.ptr(gv_in), _ffi.ptr(b["uv"]),
                      _ffi.ptr(b["prod"]), _ffi.ptr(b["a"]), _ffi.ptr(b["c"]), n, _ffi.ptr(w[u + "a/TP"]), self.act_upd,
                      0.0, _ffi.ptr(b["h2"]), _ffi.ptr(w[u + "dense1/TP"]), _ffi.ptr(self.g_zp), _ffi.ptr(self.g_uv),
                      _ffi.stream())
            self._chain(self.g_uv, 3 * n, 256, w["uvT%d/P" % i], None, 128, self.g_vp, addend=gv_in)
            if tiles:     # sender tiles in LDS, filter and its derivative on the matrix pipe; one slice of g_d / g_rij
                tl = self.tiles1
                _ffi.call("mp_painn_message_bwd_tiles_f32", _ffi.ptr(b["s"]), _ffi.ptr(v_in), n, _ffi.ptr(self.rbf),
                          _ffi.ptr(self.rbfd), self.B, _ffi.ptr(self.env), _ffi.ptr(self.envd), _ffi.ptr(self.rij),
                          _ffi.ptr(w[c + "w/F"]), _ffi.ptr(self.ptr1), _ffi.ptr(self.perm1), _ffi.ptr(self.recv), m,
                          _ffi.ptr(tl["table"]), tl["count"], tl["max_rows"], tl["max_own"], tl["max_edges"], _ffi.ptr(self.g_zp),
                          _ffi.ptr(self.g_vp), _ffi.ptr(self.g_s), _ffi.ptr(self.gv) if i > 0 else None,
                          _ffi.ptr(self.g_d), _ffi.ptr(self.g_rij), 0 if i == self.depth - 1 else 1, _ffi.stream())
            else:
                _ffi.call("mp_painn_message_bwd_f32", _ffi.ptr(b["s"]), _ffi.ptr(v_in), n, _ffi.ptr(self.rbf),
                          _ffi.ptr(self.rbfd), self.B, _ffi.ptr(self.env), _ffi.ptr(self.envd), _ffi.ptr(self.rij),
                          _ffi.ptr(p[c + "w/kernel"]), _ffi.ptr(p.get(c + "w/bias")), _ffi.ptr(self.ptr1),
                          _ffi.ptr(self.perm1), _ffi.ptr(self.recv), m, _ffi.ptr(self.g_zp), _ffi.ptr(self.g_vp),
                          _ffi.ptr(self.g_s), _ffi.ptr(self.gv) if i > 0 else None, _ffi.ptr(self.g_d),
                          _ffi.ptr(self.g_rij), 0 if i == self.depth - 1 else 1, _ffi.stream())
            if i > 0:   # block 0's inputs (embedding, constant v) do not depend on the coordinates
                self._chain(self.g_s, n, 384, w[c + "phi/TP"], None, 128, self.gz, act=self.act_conv, grad_pre=b["h1"],
                            w2=w[c + "dense1/TP"], u2=128, addend=self.g_zp)
        _ffi.call("mp_edge_geometry_bwd_f32", _ffi.ptr(self.g_d), _ffi.ptr(self.g_rij), 1 if tiles else 2, _ffi.ptr(self.rij),
                  _ffi.ptr(self.dist), _ffi.ptr(self.ptr0), _ffi.ptr(self.perm0), _ffi.ptr(self.ptr1),
                  _ffi.ptr(self.perm1), n, m, -1.0, _ffi.ptr(self.force), _ffi.stream())

    def _launch(self, with_forces):
        if self._pre is not None:
            self._pre()
        self._forward(with_forces)
        if with_forces:
            self._backward()

    # ------------------------------------------------------------------------------------------------ execution
    def run_current(self, with_forces, how="graph"):
        """One pass of the bound batch on torch's current stream.  Returns ``(energy (G', L), force (N, 3) | None)`` -
        this slot's static buffers; ``force`` is the physical force -dE/dx."""
        if with_forces and not self.grad:
            raise _ffi.EngineError("this batch was bound without the reverse-pass buffers")
        if how == "graph":
            graph = self.graphs.get(bool(with_forces))
            if graph is None:
                graph = self.graphs[bool(with_forces)] = self._capture(with_forces)
            _ffi.call("mp_graph_launch", graph, _ffi.stream())
        else:
            self._launch(with_forces)
        eng = self.post_out[-1]
        return (eng if self.out_rows == self.G else eng[:self.out_rows]), (self.force if with_forces else None)

    def _capture(self, with_forces, eng=None, force=None):
        """Capture the launches of one pass on the slot's private stream; ``eng`` / ``force``: result buffers other than
        the static ones (a result-ring entry) - the last kernels of the captured pass write there."""
        saved = (self.post_out[-1], getattr(self, "force", None))
        if eng is not None:
            self.post_out[-1] = eng
        if force is not None:
            self.force = force
        try:
            torch.cuda.current_stream().synchronize()
            with torch.cuda.stream(self.stream):
                self._launch(with_forces)   # warm-up outside capture
                self.stream.synchronize()
                _ffi.call("mp_graph_begin", _ffi.stream())
                try:
                    self._launch(with_forces)
                finally:
                    exe = ctypes.c_void_p()
                    _ffi.call("mp_graph_end", _ffi.stream(), ctypes.byref(exe))
        finally:
            self.post_out[-1] = saved[0]
            if force is not None:
                self.force = saved[1]
        return exe

    def run_graph_fresh(self, with_forces):
        """Graph replay into result tensors nobody else holds (``result_ring.ResultRing``: no copy launches): returns
        ``(energy (G', L), force (N, 3) | None)``, or ``None`` when every set of the ring is still held."""
        if with_forces and not self.grad:
            raise _ffi.EngineError("this batch was bound without the reverse-pass buffers")
        ring = self.rings.get(bool(with_forces))
        if ring is None:
            ring = self.rings[bool(with_forces)] = ResultRing()
        like = self.post_out[-1]

        def make():
            bufs = (torch.empty_like(like),)
            return bufs + ((torch.empty_like(self.force),) if with_forces else ())

        got = ring.acquire(make, lambda bufs: self._capture(with_forces, bufs[0], bufs[1] if with_forces else None))
        if got is None:
            return None
        bufs, graph = got
        _ffi.call("mp_graph_launch", graph, _ffi.stream())
        eng = bufs[0]
        return (eng if self.out_rows == self.G else eng[:self.out_rows]), (bufs[1] if with_forces else None)

    def check_flags(self):
        f = int(self.flags.item())
        if f & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")

    def __del__(self):
        try:
            for g in self.graphs.values():
                _ffi.call("mp_graph_destroy", g)
            for ring in self.rings.values():
                ring.destroy()
        except Exception:
            pass


class PainnFusedRoute:
    """The fused pipeline behind ``PAiNN.make_model(...)(inputs)`` and behind ``EnergyForceModel`` wrapping such a model.

    Same contract as ``fused.SchnetFusedRoute``: the kernels read the model's live weight tensors (derived layouts are
    refreshed when a weight's version counter moves), every distinct input set owns a batch slot, the first call issues
    the launches eagerly, later calls replay the slot's captured HIP graph, everything runs on torch's current stream,
    results are returned as fresh tensors."""

    def __init__(self, tensors, cfg, max_slots=8):
        self._tensors, self.cfg = tensors, cfg
        self.depth = int(cfg["depth"])
        self.n_out = len(_as_list(cfg["output_mlp"]["units"], 1))
        self.single_state = int(_as_list(cfg["output_mlp"]["units"], 1)[-1]) == 1
        self.max_slots = int(max_slots)
        self.mode = "auto"       # auto: eager launches on first sight of a batch, graph replay afterwards | graph | eager
        self.copy_output = True
        self._slots, self._p, self._wkey, self._images = {}, None, None, None
        self._wlist, self._vsum, self._wcalls, self._wepoch = None, 0, 0, -1
        self._groups, self.max_groups = {}, 4     # launch groups (call_group): k bound batches served by one launch sequence
        self.last = None

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
        torch.cuda.synchronize()
        if moved:
            self._slots.clear()
            self._groups.clear()
            self._p = {k: v for k, v in p.items() if v is not None}
            self._images = make_images(self._p, self.depth, self.n_out)
        else:
            make_images(self._p, self.depth, self.n_out, out=self._images)
        self._wkey = key

    @staticmethod
    def _key(node, xyz, idx, grad):
        return (node.values.data_ptr(), xyz.values.data_ptr(), idx.values.data_ptr(), node.row_splits.data_ptr(),
                idx.row_splits.data_ptr(), int(node.values.shape[0]), int(idx.values.shape[0]), node.nrows(),
                idx.values._version, idx.row_splits._version, node.row_splits._version, bool(grad))

    def _slot(self, inputs, grad):
        node, xyz, idx = inputs
        self._sync_weights()
        key = self._key(node, xyz, idx, grad)
        slot = self._slots.get(key)
        if slot is None:
            slot = FusedPainn(self._p, self._images, self.cfg)
            slot.bind(node, xyz, idx, grad)
            while len(self._slots) >= self.max_slots:
                self._slots.pop(next(iter(self._slots)))
            self._slots[key] = slot
        elif next(reversed(self._slots)) != key:
            self._slots[key] = self._slots.pop(key)
        slot.calls += 1
        how = self.mode
        if how == "auto":
            how = "eager" if slot.calls == 1 else "graph"
        self.last = how
        return slot, how

    def __call__(self, inputs):
        slot, how = self._slot(inputs, grad=False)
        if how == "graph" and self.copy_output:
            got = slot.run_graph_fresh(False)   # a result buffer nobody else holds: no copy launch
            if got is not None:
                return got[0]
        eng, _ = slot.run_current(False, how)
        return eng.clone() if self.copy_output else eng

    def energy_force(self, inputs):
        """``(energy (G', 1), force (N, 3))`` with force = -dE/dx (physical sign)."""
        if not self.single_state:
            raise ValueError("fused forces are built for one energy state")
        slot, how = self._slot(inputs, grad=True)
        if how == "graph" and self.copy_output:
            got = slot.run_graph_fresh(True)
            if got is not None:
                return got
        eng, force = slot.run_current(True, how)
        return (eng.clone(), force.clone()) if self.copy_output else (eng, force)

    def call_group(self, inputs_list, with_forces=False):
        """k independent batches from ONE launch sequence (``group.BatchUnion``: concatenated on the device by one kernel,
        then the ordinary pipeline on the union, the concatenation inside the captured graph).  Returns a list of energies
        ``(G_b', 1)`` - or of ``(energy, force (N_b, 3))`` pairs with ``with_forces`` - as views of result buffers nobody else
        holds; every member gets the rows of a call of its own (graphs do not interact)."""
        from .group import BatchUnion
        inputs_list = [list(x) for x in inputs_list]
        if len(inputs_list) == 1 or not all(self.accepts(x, with_forces=with_forces) for x in inputs_list):
            one = self.energy_force if with_forces else self.__call__
            return [one(x) for x in inputs_list]
        if with_forces and not self.single_state:
            raise ValueError("fused forces are built for one energy state")
        self._sync_weights()
        key = tuple(self._key(*x, with_forces) for x in inputs_list)
        union = self._groups.get(key)
        if union is None:
            union = BatchUnion(inputs_list)
            union.concat()                                    # the index plan of the union is made from its values at bind
            while len(self._groups) >= self.max_groups:
                self._groups.pop(next(iter(self._groups)))
            self._groups[key] = union
        elif next(reversed(self._groups)) != key:
            self._groups[key] = self._groups.pop(key)
        slot, how = self._slot(union.inputs, grad=with_forces)
        slot._pre = union.concat
        got = slot.run_graph_fresh(with_forces) if (how == "graph" and self.copy_output) else None
        if got is None:
            eng, force = slot.run_current(with_forces, how)
            got = (eng.clone(), force.clone() if with_forces else None) if self.copy_output else (eng, force)
        eng, force = got if isinstance(got, tuple) else (got, None)
        energies = union.split_graphs(eng)
        if not with_forces:
            return energies
        return list(zip(energies, union.split_nodes(force)))

    def slot_of(self, inputs, grad=False):
        return self._slots.get(self._key(*inputs, grad))

    def check_flags(self):
        for slot in self._slots.values():
            slot.check_flags()

    def release(self):
        torch.cuda.synchronize()
        self._wlist = None
        self._groups.clear()
        self._slots.clear()
