"""SchNet energy + forces from one HIP graph: fused forward kernels + a hand-written reverse pass
(kgcnn/model/force.py:159-201 around kgcnn/literature/Schnet.py:104-148; the configuration of the fork's force_schnet.py).

Forward: stage 0 (embedding chain + index pass + distances) and the fused cfconv kernel as in ``fused.FusedSchnet``; the
node side runs as GEMMs that keep their pre-activations (``mp_dense_ex_f32`` with ``out_pre``) because the reverse pass
needs ``ssp'`` of them.  Reverse, per block, last to first:

    g_pre2 = (g_n W3^T) * ssp'(pre2)          g_agg = g_pre2 W2^T                           (two GEMMs, epilogue derivative)
    g_d   += cfconv distance gradient(x_i, g_agg)          mp_cfconv_gauss_dist_grad_f32    (three MFMA chains per tile)
    g_x    = cfconv(g_agg) with the index columns swapped  mp_cfconv_gauss_fused_f32        (the forward kernel itself)
    g_n   += g_x Wx^T                                                                        (GEMM, addend epilogue)

then ``-dE/dx`` from ``g_d`` over both CSRs (``mp_edge_geometry_bwd_f32``).  Block 0's input does not depend on the
coordinates, so its ``g_x`` / ``g_n`` steps are skipped.  No tape: every saved tensor is a buffer of the batch slot.
"""
import ctypes

import numpy as np
import torch

from . import _ffi

_SSP = 2   # MP_ACT_SHIFTED_SOFTPLUS


def make_grad_images(p, depth, bins, linear_head, out=None):
    """Transposed kernels for the reverse GEMMs and the cfconv reverse images, once per weight update."""
    nfl = _ffi.lib().mp_cfconv_bwd_packed_floats()
    names = ["last_mlp/0/kernel", "last_mlp/1/kernel"]
    names += ["last_mlp/2/kernel"] if linear_head else ["output_mlp/0/kernel", "output_mlp/1/kernel"]
    for i in range(depth):
        names += ["interaction%d/dense%d/kernel" % (i, k) for k in (1, 2, 3)]
    if out is None:
        out = {"T": {k: p[k].t().contiguous().clone() for k in names},
               "cf": [torch.empty(nfl, dtype=torch.float32, device="cuda") for _ in range(depth)]}
    else:
        for k in names:
            out["T"][k].copy_(p[k].t())
    for i in range(depth):
        pre = "interaction%d/cfconv/" % i
        _ffi.call("mp_cfconv_bwd_pack_f32", _ffi.ptr(p[pre + "dense1/kernel"]), _ffi.ptr(p.get(pre + "dense1/bias")),
                  int(bins), _ffi.ptr(p[pre + "dense2/kernel"]), _ffi.ptr(out["cf"][i]), _ffi.stream())
    torch.cuda.current_stream().synchronize()
    return out


class FusedSchnetForce:
    """One batch slot of the energy + force pass."""

    def __init__(self, p, images, grad_images, depth, gauss_args, fast_softplus=True):
        self.p, self.w, self.gw = p, images, grad_images
        self.depth, self.gauss = int(depth), dict(gauss_args)
        self.flags_arg = (1 if fast_softplus else 0) | 2
        self.emb_dim = int(p["embedding"].shape[1])
        self.linear_head = "output_mlp/0/kernel" not in p
        self.stream = torch.cuda.Stream()
        self.graph = None
        self.calls = 0

    def bind(self, node, xyz, idx):
        self.inputs = (node, xyz, idx)
        n, m, g = int(node.values.shape[0]), int(idx.values.shape[0]), node.nrows()
        self.N, self.M, self.G = n, m, g
        dev, f32 = node.values.device, torch.float32
        e = lambda *shape: torch.empty(shape, dtype=f32, device=dev)
        plan = idx.index_plan(node)
        if plan.flags_host() & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")
        self.ptr0, self.perm0, self.recv_sorted = plan.csr(0)
        self.ptr1, self.perm1, self.send_sorted = plan.csr(1)
        self.node_flags = (self.flags_arg & 3) | (256 if node.values.dtype == torch.int64 else 0)
        mm = max(m, 1)
        self.recv = torch.empty(mm, dtype=torch.int32, device=dev)
        self.send = torch.empty(mm, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self.dist, self.rij = e(mm), e(mm, 3)
        d = self.depth
        self.zero_pool = torch.zeros((2 * d, n, 128), dtype=f32, device=dev)   # agg_i and g_x_i: zero on entry
        self.n = [e(n, 128) for _ in range(2)]
        self.xs = [e(n, 128) for _ in range(d)]
        self.pre2 = [e(n, 128) for _ in range(d)]
        self.t = e(n, 128)
        self.pl0, self.u0, self.pl1, self.h = e(n, 128), e(n, 128), e(n, 64), e(n, 64)
        self.energy = e(g, 1)
        if self.linear_head:
            self.y = e(n, 1)
        else:
            self.pooled, self.po0, self.o0 = e(g, 64), e(g, 64), e(g, 64)
        splits = node.row_splits_host()
        rows = g
        while rows > 0 and splits[rows] == splits[rows - 1]:
            rows -= 1
        self.out_rows = rows
        # reverse pass
        self.ones = torch.ones((g, 1), dtype=f32, device=dev)
        self.g_small = [e(g, 64), e(g, 64)]
        self.g_h, self.g_pl1, self.g_pl0 = e(n, 64), e(n, 64), e(n, 128)
        self.g_y = e(n, 1)
        self.g_n, self.g_pre2, self.g_agg = e(n, 128), e(n, 128), e(n, 128)
        self.g_d = e(mm)
        self.g_rij0 = torch.zeros((mm, 3), dtype=f32, device=dev)   # SchNet has no direction-dependent term
        self.force = e(n, 3)
        self.graph = None

    @staticmethod
    def _dense(x, rows, k, w, b, u, out, act=0, out_pre=None, grad_act=0, grad_pre=None, addend=None):
        _ffi.call("mp_dense_ex_f32", _ffi.ptr(x), rows, k, _ffi.ptr(w), _ffi.ptr(b), u, act, 0.0, 0, grad_act, 0.0,
                  None, _ffi.ptr(addend), _ffi.ptr(out_pre), _ffi.ptr(grad_pre), _ffi.ptr(out), _ffi.stream())

    def _cfconv(self, x, packed, out, swapped=False):
        ga = self.gauss
        if not swapped:   # receiver-sorted list (the stable-sort permutation if the batch is not sorted)
            seg, other, perm = (self.recv if self.perm0 is None else self.recv_sorted), self.send, self.perm0
        else:             # sender-sorted list: out[send] += x[recv] * w
            seg, other, perm = (self.send if self.perm1 is None else self.send_sorted), self.recv, self.perm1
        _ffi.call("mp_cfconv_gauss_fused_f32", _ffi.ptr(x), self.N, _ffi.ptr(self.dist), int(ga["bins"]),
                  float(ga["distance"]), float(ga["sigma"]), float(ga["offset"]), _ffi.ptr(packed), _ffi.ptr(seg),
                  _ffi.ptr(other), _ffi.ptr(perm), self.M, self.flags_arg, _ffi.ptr(out), _ffi.stream())

    def _launch(self):
        p, w, gw, n, m, g, d = self.p, self.w, self.gw, self.N, self.M, self.G, self.depth
        node, xyz, idx = self.inputs
        nimg, T = w["node"], gw["T"]
        ga = self.gauss
        # ---------------------------------------------------------------------------------------------- forward
        self.zero_pool.zero_()
        _ffi.call("mp_schnet_stage0_f32", _ffi.ptr(node.values), n, _ffi.ptr(p["embedding"]),
                  int(p["embedding"].shape[0]), self.emb_dim, _ffi.ptr(nimg["dense0/kernel"]),
                  _ffi.ptr(p.get("dense0/bias")), _ffi.ptr(nimg["interaction0/dense1/kernel"]), _ffi.ptr(self.n[0]),
                  _ffi.ptr(self.xs[0]), _ffi.ptr(idx.values), m, _ffi.ptr(node.row_splits), _ffi.ptr(idx.row_splits), g,
                  _ffi.ptr(xyz.values), _ffi.ptr(self.recv), _ffi.ptr(self.send), _ffi.ptr(self.dist),
                  _ffi.ptr(self.flags), self.node_flags, _ffi.stream())
        if m > 0:
            _ffi.call("mp_edge_geometry_f32", _ffi.ptr(xyz.values), n, _ffi.ptr(self.recv), _ffi.ptr(self.send), m, None,
                      _ffi.ptr(self.rij), _ffi.stream())
        cur = 0
        for i in range(d):
            pre = "interaction%d/" % i
            agg = self.zero_pool[i]
            self._cfconv(self.xs[i], w["cfconv"][i], agg)
            self._dense(agg, n, 128, p[pre + "dense2/kernel"], p.get(pre + "dense2/bias"), 128, self.t, act=_SSP,
                        out_pre=self.pre2[i])
            self._dense(self.t, n, 128, p[pre + "dense3/kernel"], p.get(pre + "dense3/bias"), 128, self.n[1 - cur],
                        addend=self.n[cur])
            cur = 1 - cur
            if i + 1 < d:
                self._dense(self.n[cur], n, 128, p["interaction%d/dense1/kernel" % (i + 1)], None, 128, self.xs[i + 1])
        self._dense(self.n[cur], n, 128, p["last_mlp/0/kernel"], p.get("last_mlp/0/bias"), 128, self.u0, act=_SSP,
                    out_pre=self.pl0)
        self._dense(self.u0, n, 128, p["last_mlp/1/kernel"], p.get("last_mlp/1/bias"), 64, self.h, act=_SSP,
                    out_pre=self.pl1)
        if self.linear_head:
            self._dense(self.h, n, 64, p["last_mlp/2/kernel"], p.get("last_mlp/2/bias"), 1, self.y)
            _ffi.call("mp_pool_graph_f32", _ffi.MP_SUM, _ffi.ptr(self.y), _ffi.ptr(node.row_splits), g, 1, None,
                      _ffi.ptr(self.energy), _ffi.stream())
        else:
            _ffi.call("mp_pool_graph_f32", _ffi.MP_SUM, _ffi.ptr(self.h), _ffi.ptr(node.row_splits), g, 64, None,
                      _ffi.ptr(self.pooled), _ffi.stream())
            self._dense(self.pooled, g, 64, p["output_mlp/0/kernel"], p.get("output_mlp/0/bias"), 64, self.o0, act=_SSP,
                        out_pre=self.po0)
            self._dense(self.o0, g, 64, p["output_mlp/1/kernel"], p.get("output_mlp/1/bias"), 1, self.energy)
        # ---------------------------------------------------------------------------------------------- reverse
        if self.linear_head:
            _ffi.call("mp_repeat_rows_f32", _ffi.ptr(self.ones), _ffi.ptr(node.row_splits), g, 1, n, _ffi.ptr(self.g_y),
                      _ffi.stream())
            self._dense(self.g_y, n, 1, T["last_mlp/2/kernel"], None, 64, self.g_pl1, grad_act=_SSP, grad_pre=self.pl1)
        else:
            self._dense(self.ones, g, 1, T["output_mlp/1/kernel"], None, 64, self.g_small[0], grad_act=_SSP,
                        grad_pre=self.po0)
            self._dense(self.g_small[0], g, 64, T["output_mlp/0/kernel"], None, 64, self.g_small[1])
            _ffi.call("mp_repeat_rows_f32", _ffi.ptr(self.g_small[1]), _ffi.ptr(node.row_splits), g, 64, n,
                      _ffi.ptr(self.g_h), _ffi.stream())
            _ffi.call("mp_activation_grad_f32", _SSP, 0.0, _ffi.ptr(self.pl1), _ffi.ptr(self.g_h), n * 64,
                      _ffi.ptr(self.g_pl1), _ffi.stream())
        self._dense(self.g_pl1, n, 64, T["last_mlp/1/kernel"], None, 128, self.g_pl0, grad_act=_SSP, grad_pre=self.pl0)
        self._dense(self.g_pl0, n, 128, T["last_mlp/0/kernel"], None, 128, self.g_n)
        for i in range(d - 1, -1, -1):
            pre = "interaction%d/" % i
            self._dense(self.g_n, n, 128, T[pre + "dense3/kernel"], None, 128, self.g_pre2, grad_act=_SSP,
                        grad_pre=self.pre2[i])
            self._dense(self.g_pre2, n, 128, T[pre + "dense2/kernel"], None, 128, self.g_agg)
            _ffi.call("mp_cfconv_gauss_dist_grad_f32", _ffi.ptr(self.xs[i]), _ffi.ptr(self.g_agg), n, _ffi.ptr(self.dist),
                      int(ga["bins"]), float(ga["distance"]), float(ga["sigma"]), float(ga["offset"]),
                      _ffi.ptr(gw["cf"][i]), _ffi.ptr(self.recv), _ffi.ptr(self.send), m, 0 if i == d - 1 else 1,
                      _ffi.ptr(self.g_d), _ffi.stream())
            if i > 0:   # block 0's x comes from the embedding: no path to the coordinates
                g_x = self.zero_pool[d + i]
                self._cfconv(self.g_agg, w["cfconv"][i], g_x, swapped=True)
                self._dense(g_x, n, 128, T[pre + "dense1/kernel"], None, 128, self.g_n, addend=self.g_n)
        _ffi.call("mp_edge_geometry_bwd_f32", _ffi.ptr(self.g_d), _ffi.ptr(self.g_rij0), 1, _ffi.ptr(self.rij),
                  _ffi.ptr(self.dist), _ffi.ptr(self.ptr0), _ffi.ptr(self.perm0), _ffi.ptr(self.ptr1),
                  _ffi.ptr(self.perm1), n, m, -1.0, _ffi.ptr(self.force), _ffi.stream())

    def run_current(self, how="graph"):
        """Energy ``(G', 1)`` and physical force ``(N, 3)`` of the bound batch on torch's current stream (static buffers)."""
        if how == "graph":
            if self.graph is None:
                torch.cuda.current_stream().synchronize()
                with torch.cuda.stream(self.stream):
                    self._launch()
                    self.stream.synchronize()
                    _ffi.call("mp_graph_begin", _ffi.stream())
                    try:
                        self._launch()
                    finally:
                        exe = ctypes.c_void_p()
                        _ffi.call("mp_graph_end", _ffi.stream(), ctypes.byref(exe))
                self.graph = exe
            _ffi.call("mp_graph_launch", self.graph, _ffi.stream())
        else:
            self._launch()
        eng = self.energy if self.out_rows == self.G else self.energy[:self.out_rows]
        return eng, self.force

    def check_flags(self):
        if int(self.flags.item()) & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")

    def __del__(self):
        try:
            if self.graph is not None:
                _ffi.call("mp_graph_destroy", self.graph)
        except Exception:
            pass
