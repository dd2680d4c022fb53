"""SchNet energy + forces from one HIP graph: fused forward kernels + a hand-written reverse pass
(kgcnn/model/force.py:159-201 around kgcnn/literature/Schnet.py:104-148; the configuration of the fork's force_schnet.py).

One C-ABI call, ``mp_schnet_force_launch`` (csrc/mp_schnet_bwd.hip), issues the whole pass from a descriptor of the bound
batch slot:

    forward   stage 0, per block cfconv + node chain - the kernels of ``fused.FusedSchnet`` in their SAVE builds, which keep
              sigmoid(pre-activation) of every shifted softplus and the sender features x_i of every block - and the
              readout (which also writes dE/d pooled for the MLP head)
    reverse   head chain (5 GEMMs, one launch), then per block, last to first:
                  g_d  += cfconv distance gradient(x_i, g_agg)      mp_cfconv_gauss_dist_grad_f32
                  g_x   = cfconv(g_agg) with the index columns swapped (the forward kernel itself)
                  block chain: g_n += g_x Wx^T ; g_agg = ((g_n W3^T) * d2) W2^T     (3 GEMMs, one launch)
              and -dE/dx from g_d over both CSRs (``mp_schnet_force_from_gd_f32``).

Block 0's input does not depend on the coordinates, so its ``g_x`` step is skipped.  No tape: every saved tensor is a
buffer of the batch slot.  32 launches for depth 6 (the first version - one GEMM per launch - had 75 and took twice as long).
"""
import ctypes

import torch

from . import _ffi
from .result_ring import ResultRing


def _transposed_names(depth, linear_head):
    names = ["last_mlp/0/kernel", "last_mlp/1/kernel"]
    for i in range(depth):
        names += ["interaction%d/dense%d/kernel" % (i, k) for k in ((2, 3) if i == 0 else (1, 2, 3))]
    return names


def make_grad_images(p, depth, bins, linear_head, out=None):
    """Images for the reverse pass, once per weight update: ``mp_schnet_node_pack_f32`` images of the TRANSPOSED node-side
    kernels (``"T"``) and the cfconv reverse images (``"cf"``, ``mp_cfconv_bwd_pack_f32``).  ``out`` re-fills in place."""
    nfl = _ffi.lib().mp_cfconv_bwd_packed_floats()
    names = _transposed_names(depth, linear_head)
    if out is None:
        out = {"T": {k: torch.empty(p[k].numel(), dtype=torch.float32, device="cuda") for k in names},
               "cf": [torch.empty(nfl, dtype=torch.float32, device="cuda") for _ in range(depth)]}
    for k in names:
        wt = p[k].t().contiguous()
        _ffi.call("mp_schnet_node_pack_f32", _ffi.ptr(wt), int(wt.shape[0]), int(wt.shape[1]), _ffi.ptr(out["T"][k]),
                  _ffi.stream())
        torch.cuda.current_stream().synchronize()   # wt is a temporary
    for i in range(depth):
        pre = "interaction%d/cfconv/" % i
        _ffi.call("mp_cfconv_bwd_pack_f32", _ffi.ptr(p[pre + "dense1/kernel"]), _ffi.ptr(p.get(pre + "dense1/bias")),
                  int(bins), _ffi.ptr(p[pre + "dense2/kernel"]), _ffi.ptr(out["cf"][i]), _ffi.stream())
    torch.cuda.current_stream().synchronize()
    return out


class FusedSchnetForce:
    """One batch slot of the energy + force pass."""

    def __init__(self, p, images, grad_images, depth, gauss_args, fast_softplus=True, cfconv_flags=0):
        self.p, self.w, self.gw = p, images, grad_images
        self.depth, self.gauss = int(depth), dict(gauss_args)
        if self.depth > _ffi.MP_SCHNET_MAX_DEPTH:
            raise ValueError("fused SchNet force pass: depth <= %d" % _ffi.MP_SCHNET_MAX_DEPTH)
        self.flags_arg = (1 if fast_softplus else 0) | 2 | int(cfconv_flags)
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
        z = lambda *shape: torch.zeros(shape, dtype=f32, device=dev)
        plan = idx.index_plan(node)
        if plan.flags_host() & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")
        self.ptr0, self.perm0, self.seg0 = plan.csr(0)
        self.ptr1, self.perm1, self.seg1 = plan.csr(1)
        mm, d = max(m, 1), self.depth
        self.recv = torch.empty(mm, dtype=torch.int32, device=dev)
        self.send = torch.empty(mm, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self.dist = e(mm)
        self.n, self.agg, self.h, self.energy = e(n, 128), z(n, 128), e(n, 64), e(g, 1)
        self.xs, self.d2 = e(d, n, 128), e(d, n, 128)
        self.dl0, self.dl1 = e(n, 128), e(n, 64)
        self.g_n, self.g_agg, self.g_x, self.g_d = e(n, 128), e(n, 128), z(n, 128), z(mm)
        self.force = e(n, 3)
        self.g_pool = self.node_graph = None
        if not self.linear_head:
            self.g_pool = e(g, 64)
            counts = node.row_splits[1:] - node.row_splits[:-1]
            self.node_graph = torch.repeat_interleave(torch.arange(g, dtype=torch.int32, device=dev), counts,
                                                      output_size=n).contiguous()
        splits = node.row_splits_host()
        rows = g
        while rows > 0 and splits[rows] == splits[rows - 1]:
            rows -= 1
        self.out_rows = rows
        self._desc = self._descriptor()
        self._desc_ref = ctypes.byref(self._desc)
        self._launch_fn = _ffi.lib().mp_schnet_force_launch
        self._drop_graphs()
        self._ring = ResultRing()

    def _descriptor(self, energy=None, force=None):
        """``mp_schnet_force_desc`` of the bound batch; ``energy`` / ``force``: result buffers other than the static
        ones (a result-ring entry)."""
        energy = self.energy if energy is None else energy
        force = self.force if force is None else force
        p, w, gw, ga = self.p, self.w["node"], self.gw, self.gauss
        node, xyz, idx = self.inputs
        f = _ffi.SchnetForceDesc()
        d = f.fwd
        addr = lambda t: None if t is None else t.data_ptr()
        d.N, d.M, d.G = self.N, self.M, self.G
        d.depth, d.vocab, d.bins = self.depth, int(p["embedding"].shape[0]), int(ga["bins"])
        d.flags = self.flags_arg | (256 if node.values.dtype == torch.int64 else 0)
        d.g_distance, d.g_sigma, d.g_offset = float(ga["distance"]), float(ga["sigma"]), float(ga["offset"])
        d.emb_dim = self.emb_dim
        d.numbers, d.xyz, d.idx = addr(node.values), addr(xyz.values), addr(idx.values)
        d.node_splits, d.edge_splits = addr(node.row_splits), addr(idx.row_splits)
        d.embedding, d.W0, d.b0 = addr(p["embedding"]), addr(w["dense0/kernel"]), addr(p.get("dense0/bias"))
        T = gw["T"]
        for i in range(self.depth):
            pre = "interaction%d/" % i
            d.Wx[i], d.packed[i] = addr(w[pre + "dense1/kernel"]), addr(self.w["cfconv"][i])
            d.W2[i], d.b2[i] = addr(w[pre + "dense2/kernel"]), addr(p.get(pre + "dense2/bias"))
            d.W3[i], d.b3[i] = addr(w[pre + "dense3/kernel"]), addr(p.get(pre + "dense3/bias"))
            f.W3T[i], f.W2T[i] = addr(T[pre + "dense3/kernel"]), addr(T[pre + "dense2/kernel"])
            f.WxT[i] = addr(T.get(pre + "dense1/kernel"))
            f.packed_bwd[i] = addr(gw["cf"][i])
        d.Wl0, d.bl0 = addr(w["last_mlp/0/kernel"]), addr(p.get("last_mlp/0/bias"))
        d.Wl1, d.bl1 = addr(w["last_mlp/1/kernel"]), addr(p.get("last_mlp/1/bias"))
        if self.linear_head:
            d.Wo0, d.bo0, d.Wo1, d.bo1 = None, None, addr(p["last_mlp/2/kernel"]), addr(p.get("last_mlp/2/bias"))
        else:
            d.Wo0, d.bo0 = addr(p["output_mlp/0/kernel"]), addr(p.get("output_mlp/0/bias"))
            d.Wo1, d.bo1 = addr(p["output_mlp/1/kernel"]), addr(p.get("output_mlp/1/bias"))
        d.recv, d.send, d.dist, d.flags_word = addr(self.recv), addr(self.send), addr(self.dist), addr(self.flags)
        d.n, d.x, d.agg, d.h, d.out = addr(self.n), None, addr(self.agg), addr(self.h), addr(energy)
        f.xs, f.d2, f.dl0, f.dl1 = addr(self.xs), addr(self.d2), addr(self.dl0), addr(self.dl1)
        f.g_pool, f.node_graph = addr(self.g_pool), addr(self.node_graph)
        f.Wl0T, f.Wl1T = addr(T["last_mlp/0/kernel"]), addr(T["last_mlp/1/kernel"])
        # a column that is already sorted has no permutation: the kernels then read the column stage 0 writes
        f.seg0, f.perm0 = (None, None) if self.perm0 is None else (addr(self.seg0), addr(self.perm0))
        f.seg1, f.perm1 = (None, None) if self.perm1 is None else (addr(self.seg1), addr(self.perm1))
        f.ptr0, f.ptr1 = addr(self.ptr0), addr(self.ptr1)
        f.g_n, f.g_agg, f.g_x, f.g_d, f.force = (addr(self.g_n), addr(self.g_agg), addr(self.g_x), addr(self.g_d),
                                                  addr(force))
        f.force_scale = -1.0
        return f

    def _launch(self, desc_ref=None):
        _ffi.check(self._launch_fn(self._desc_ref if desc_ref is None else desc_ref,
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def _capture(self, desc_ref=None):
        torch.cuda.current_stream().synchronize()
        with torch.cuda.stream(self.stream):
            self._launch(desc_ref)
            self.stream.synchronize()
            _ffi.call("mp_graph_begin", _ffi.stream())
            try:
                self._launch(desc_ref)
            finally:
                exe = ctypes.c_void_p()
                _ffi.call("mp_graph_end", _ffi.stream(), ctypes.byref(exe))
        return exe

    def _capture_into(self, bufs):
        desc = self._descriptor(bufs[0], bufs[1])
        return self._capture(ctypes.byref(desc))   # the descriptor is read at launch (capture) time only

    def run_current(self, how="graph"):
        """Energy ``(G', 1)`` and physical force ``(N, 3)`` of the bound batch on torch's current stream (static buffers)."""
        if how == "graph":
            if self.graph is None:
                self.graph = self._capture()
            _ffi.call("mp_graph_launch", self.graph, _ffi.stream())
        else:
            self._launch()
        eng = self.energy if self.out_rows == self.G else self.energy[:self.out_rows]
        return eng, self.force

    def run_graph_fresh(self):
        """Graph replay into an (energy, force) pair nobody else holds (``result_ring.ResultRing``: no copy launches), or
        ``None`` when every pair of the ring is still held."""
        dev = self.energy.device
        got = self._ring.acquire(lambda: (torch.empty((self.G, 1), dtype=torch.float32, device=dev),
                                          torch.empty((self.N, 3), dtype=torch.float32, device=dev)),
                                 self._capture_into)
        if got is None:
            return None
        (eng, force), graph = got
        _ffi.call("mp_graph_launch", graph, _ffi.stream())
        return (eng if self.out_rows == self.G else eng[:self.out_rows]), force

    def _drop_graphs(self):
        if getattr(self, "graph", None) is not None:
            try:
                _ffi.call("mp_graph_destroy", self.graph)
            except Exception:
                pass
        if getattr(self, "_ring", None) is not None:
            self._ring.destroy()
        self.graph = None

    def check_flags(self):
        if int(self.flags.item()) & _ffi.MP_FLAG_OOB:
            raise IndexError("edge index out of range for its graph")

    def __del__(self):
        try:
            self._drop_graphs()
        except Exception:
            pass
