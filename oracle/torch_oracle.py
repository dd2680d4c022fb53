"""torch-CPU restatement of the reference's UNFUSED SchNet op sequence (second CPU baseline leg).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: imported only by tests/ (checked against oracle/kgcnn_oracle.py) and by the
``cpu_baseline`` leg of bench.py.  Parity status: as kgcnn_oracle.py - pinned by the reference's five known answers
through the NumPy oracle this file is tested against, "parity unpinned" w.r.t. TensorFlow (not installable here).

Every step materialises its output like the TF graph of the reference does, with torch's multi-threaded CPU kernels
standing in for TF's Eigen pool: ``partition_row_indexing`` (kgcnn/ops/partition.py:140-155) -> ``index_select`` for
``tf.gather`` (kgcnn/layers/gather.py:228) -> ``addmm`` for Dense (kgcnn/layers/modules.py:85) -> multiply
(modules.py:301) -> stable argsort + gather by order (kgcnn/layers/pooling.py:66-68) -> ``index_add_`` on the sorted ids
for the sorted segment_sum with its zero pad (pooling.py:69-76), wired as kgcnn/layers/conv/schnet_conv.py:73-79,159-165
and kgcnn/literature/Schnet.py:104-148.
"""
import math

import numpy as np
import torch

_LN2 = math.log(2.0)


def _ssp(x):
    """kgcnn/ops/activ.py:15: softplus(x) - log(2) (torch's softplus thresholds at 20, TF's at ~13.9: same to 1e-9)."""
    return torch.nn.functional.softplus(x) - _LN2


def to_torch(params):
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in params.items()}


def prepare(batch):
    """Host arrays -> CPU torch tensors (outside the timed region, like the GPU run's resident inputs)."""
    return {"z": torch.from_numpy(np.ascontiguousarray(batch["node_number"], np.float32)),
            "xyz": torch.from_numpy(np.ascontiguousarray(batch["node_coordinates"], np.float32)),
            "idx": torch.from_numpy(np.ascontiguousarray(batch["edge_indices"], np.int64)),
            "ns": torch.from_numpy(np.ascontiguousarray(batch["node_splits"], np.int64)),
            "es": torch.from_numpy(np.ascontiguousarray(batch["edge_splits"], np.int64))}


def _shift(idx, ns, es):
    g = ns.numel() - 1
    graph_of_edge = torch.repeat_interleave(torch.arange(g), es[1:] - es[:-1])
    return idx + ns[:-1].index_select(0, graph_of_edge).unsqueeze(1)


def _pool_sum(edges, recv, n_rows):
    order = torch.argsort(recv, stable=True)
    srt = edges.index_select(0, order)
    seg = recv.index_select(0, order)
    return torch.zeros((n_rows, edges.shape[1]), dtype=edges.dtype).index_add_(0, seg, srt)


def schnet_forward(p, t, depth=3, gauss_args=None):
    """Same contract as ``kgcnn_oracle.schnet_forward`` (graph output ``(G, 1)``); ``p`` from :func:`to_torch`,
    ``t`` from :func:`prepare`."""
    ga = gauss_args or {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4}
    z, xyz, idx, ns, es = t["z"], t["xyz"], t["idx"], t["ns"], t["es"]
    n_rows, g = int(z.shape[0]), ns.numel() - 1
    with torch.no_grad():
        n = p["embedding"].index_select(0, z.to(torch.int32).to(torch.int64))
        sh = _shift(idx, ns, es)
        d = (xyz.index_select(0, sh[:, 0]) - xyz.index_select(0, sh[:, 1])).square().sum(-1, keepdim=True)
        d = d.clamp_min(0.0).sqrt()
        bins = int(ga["bins"])
        mu = torch.arange(bins, dtype=torch.float32) / float(bins) * float(ga["distance"])
        gamma = 1.0 / float(ga["sigma"]) / float(ga["sigma"]) / 2.0
        rbf = torch.exp(((d - float(ga["offset"])) - mu).square() * (-gamma))
        n = torch.addmm(p["dense0/bias"], n, p["dense0/kernel"])
        for i in range(depth):
            pre = "interaction%d/" % i
            x = n @ p[pre + "dense1/kernel"]
            h = _ssp(torch.addmm(p[pre + "cfconv/dense1/bias"], rbf, p[pre + "cfconv/dense1/kernel"]))
            w = torch.addmm(p[pre + "cfconv/dense2/bias"], h, p[pre + "cfconv/dense2/kernel"])
            xj = x.index_select(0, _shift(idx, ns, es)[:, 1])     # every gather / pooling call recomputes the shift
            agg = _pool_sum(xj * w, _shift(idx, ns, es)[:, 0], n_rows)
            u = _ssp(torch.addmm(p[pre + "dense2/bias"], agg, p[pre + "dense2/kernel"]))
            n = n + torch.addmm(p[pre + "dense3/bias"], u, p[pre + "dense3/kernel"])
        hl = _ssp(torch.addmm(p["last_mlp/0/bias"], n, p["last_mlp/0/kernel"]))
        hl = _ssp(torch.addmm(p["last_mlp/1/bias"], hl, p["last_mlp/1/kernel"]))
        graph_of_node = torch.repeat_interleave(torch.arange(g), ns[1:] - ns[:-1])
        pooled = torch.zeros((g, hl.shape[1]), dtype=hl.dtype).index_add_(0, graph_of_node, hl)
        o = _ssp(torch.addmm(p["output_mlp/0/bias"], pooled, p["output_mlp/0/kernel"]))
        return torch.addmm(p["output_mlp/1/bias"], o, p["output_mlp/1/kernel"]).numpy()
