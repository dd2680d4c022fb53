"""Shared builders for the GPU tests (weights in constructor order, device ragged tensors, finite differences)."""
import numpy as np


def dev(values, splits):
    from gcnn_keras_amd.ragged import RaggedTensor
    return RaggedTensor.from_numpy(values, splits)


def mol_inputs(b):
    return [dev(b["node_number"], b["node_splits"]), dev(b["node_coordinates"], b["node_splits"]),
            dev(b["edge_indices"], b["edge_splits"])]


def painn_weight_list(p, depth=3, n_out=2):
    """``synth.painn_params`` in the constructor order of ``PAiNN.make_model`` (what ``set_weights`` expects)."""
    order = ["embedding", "bessel/frequencies"]
    for i in range(depth):
        order += ["conv%d/dense1/kernel" % i, "conv%d/dense1/bias" % i, "conv%d/phi/kernel" % i, "conv%d/phi/bias" % i,
                  "conv%d/w/kernel" % i, "conv%d/w/bias" % i,
                  "update%d/dense1/kernel" % i, "update%d/dense1/bias" % i, "update%d/lin_u/kernel" % i,
                  "update%d/lin_v/kernel" % i, "update%d/a/kernel" % i, "update%d/a/bias" % i]
    for k in range(n_out):
        order += ["output_mlp/%d/kernel" % k, "output_mlp/%d/bias" % k]
    return [p[k] for k in order]


def fd_gradient(fn, x, h=1e-5):
    """Central finite differences of the scalar ``fn(x).sum()`` w.r.t. every entry of ``x`` (float64)."""
    base = np.asarray(x, dtype=np.float64)
    g = np.zeros_like(base)
    flat, gf = base.reshape(-1), g.reshape(-1)
    for i in range(flat.size):
        keep = flat[i]
        flat[i] = keep + h
        up = float(np.sum(fn(base)))
        flat[i] = keep - h
        dn = float(np.sum(fn(base)))
        flat[i] = keep
        gf[i] = (up - dn) / (2 * h)
    return g
