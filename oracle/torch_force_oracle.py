"""Analytic force reference: torch-CPU autograd restatement of ``EnergyForceModel`` (kgcnn/model/force.py:136-201).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: imported only by tests/ (and by scripts that print parity figures).  Parity status:
the energies of this file are tested against oracle/kgcnn_oracle.py (NumPy, pinned to the reference's known answers where
the reference holds any); forces are "parity unpinned" w.r.t. TensorFlow (not installable here) - they are the exact
derivative, by reverse-mode differentiation in float64 or float32, of a restatement whose forward values are checked op
for op, and are themselves checked against central finite differences of the NumPy oracle (tests/test_force_oracle.py).

The reference obtains forces as ``tape.batch_jacobian(energy, x_pad)`` (force.py:159-176): the derivative of every
state of every molecule's energy w.r.t. that molecule's padded coordinates.  Molecules of a disjoint batch do not interact
(kgcnn/ops/partition.py:140-155 shifts every graph's indices into its own node range), so the batch Jacobian of state s
equals the gradient of ``sum_b E[b, s]`` w.r.t. the flat ``(N, 3)`` coordinate values - one reverse pass per state.

Energy models restated here (differentiable, dtype = float32 or float64):
  * ``schnet_energy``: kgcnn/literature/Schnet.py:104-148 (any depth / Gauss basis / head layout, number or attribute input)
  * ``painn_energy``:  kgcnn/literature/PAiNN.py:100-155 with kgcnn/layers/conv/painn_conv.py:97-115, 201-214
"""
import math

import numpy as np
import torch

_LN2 = math.log(2.0)


def _ssp(x):
    """kgcnn/ops/activ.py:15.  ``softplus`` as log1p(exp(x)) up to x = 50 (exact in both dtypes; TF switches to ``x`` at
    ~13.9 / ~34, where the two forms agree to the last bit of the dtype)."""
    return torch.nn.functional.softplus(x, beta=1.0, threshold=50.0) - _LN2


def _swish(x):
    return x * torch.sigmoid(x)


_ACT = {None: lambda x: x, "linear": lambda x: x, "kgcnn>shifted_softplus": _ssp, "shifted_softplus": _ssp,
        "swish": _swish, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}


def _dense(x, p, name, act=None):
    y = x @ p[name + "/kernel"]
    if name + "/bias" in p:
        y = y + p[name + "/bias"]
    return _ACT[act](y)


def to_torch(params, dtype=torch.float64):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in params.items()}


def _shift(idx, ns, es):
    """``partition_row_indexing`` sample -> batch, kgcnn/ops/partition.py:140-155."""
    ns, es = torch.as_tensor(np.asarray(ns, np.int64)), torch.as_tensor(np.asarray(es, np.int64))
    graph_of_edge = torch.repeat_interleave(torch.arange(ns.numel() - 1), es[1:] - es[:-1])
    return torch.as_tensor(np.asarray(idx, np.int64)) + ns[:-1].index_select(0, graph_of_edge).unsqueeze(1)


def _pool_nodes(x, ns):
    ns = torch.as_tensor(np.asarray(ns, np.int64))
    g = ns.numel() - 1
    graph_of_node = torch.repeat_interleave(torch.arange(g), ns[1:] - ns[:-1])
    return torch.zeros((g,) + tuple(x.shape[1:]), dtype=x.dtype).index_add_(0, graph_of_node, x)


def _segment_sum(x, recv, n_rows):
    """``PoolingLocalEdges(sum)``, kgcnn/layers/pooling.py:63-78 (order of the additions is torch's: differences from the
    sequential order are rounding only)."""
    return torch.zeros((n_rows,) + tuple(x.shape[1:]), dtype=x.dtype).index_add_(0, recv, x)


def _distance(xyz, sh):
    """``NodePosition`` + ``NodeDistanceEuclidean``, kgcnn/layers/geom.py:285-327: sqrt(relu(sum (x_i - x_j)^2))."""
    diff = xyz.index_select(0, sh[:, 0]) - xyz.index_select(0, sh[:, 1])
    return diff, torch.sqrt(torch.relu(diff.square().sum(-1, keepdim=True)))


def schnet_energy(p, node_input, xyz, idx, ns, es, depth=3, gauss_args=None,
                  last_mlp_act=("kgcnn>shifted_softplus", "kgcnn>shifted_softplus"),
                  output_mlp_act=("kgcnn>shifted_softplus", "linear")):
    """Differentiable in ``xyz`` and (attribute input) in ``node_input``.  ``node_input``: node numbers ``(N,)`` when
    ``p`` holds an embedding table, otherwise 2-D attributes ``(N, k)`` (OptionalInputEmbedding, modules.py:526-534)."""
    ga = gauss_args or {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4}
    dt = xyz.dtype
    n_rows = int(xyz.shape[0])
    if "embedding" in p:
        n = p["embedding"].index_select(0, torch.as_tensor(np.asarray(node_input)).to(torch.int64))
    else:
        n = node_input
    sh = _shift(idx, ns, es)
    _, d = _distance(xyz, sh)
    bins = int(ga["bins"])
    mu = torch.arange(bins, dtype=dt) / float(bins) * float(ga["distance"])          # geom.py:554-571
    gamma = 1.0 / float(ga["sigma"]) / float(ga["sigma"]) / 2.0
    rbf = torch.exp(((d - float(ga["offset"])) - mu).square() * (-gamma))
    n = _dense(n, p, "dense0")
    for i in range(depth):
        pre = "interaction%d/" % i
        x = n @ p[pre + "dense1/kernel"]                                              # schnet_conv.py:159-165
        w = _dense(_dense(rbf, p, pre + "cfconv/dense1", "shifted_softplus"), p, pre + "cfconv/dense2")
        agg = _segment_sum(x.index_select(0, sh[:, 1]) * w, sh[:, 0], n_rows)        # schnet_conv.py:73-79
        n = n + _dense(_dense(agg, p, pre + "dense2", "shifted_softplus"), p, pre + "dense3")
    for k, act in enumerate(last_mlp_act):
        n = _dense(n, p, "last_mlp/%d" % k, act)
    out = _pool_nodes(n, ns)
    for k, act in enumerate(output_mlp_act):
        out = _dense(out, p, "output_mlp/%d" % k, act)
    return out


def painn_energy(p, node_number, xyz, idx, ns, es, depth=3, bessel_args=None, cutoff=None, equiv_method="zeros",
                 epsilon=1e-7, output_mlp_act=("swish", "linear")):
    """kgcnn/literature/PAiNN.py:100-155, graph output, sum pooling, swish."""
    ba = bessel_args or {"num_radial": 20, "cutoff": 5.0, "envelope_exponent": 5}
    dt = xyz.dtype
    n_rows = int(xyz.shape[0])
    z = p["embedding"].index_select(0, torch.as_tensor(np.asarray(node_number)).to(torch.int64))
    if equiv_method == "zeros":                                                       # painn_conv.py:261-290
        v = torch.zeros((n_rows, 3, z.shape[1]), dtype=dt)
    elif equiv_method == "eps":
        v = torch.zeros((n_rows, 3, z.shape[1]), dtype=dt) + torch.tensor(epsilon, dtype=dt)
    else:
        raise ValueError(equiv_method)
    sh = _shift(idx, ns, es)
    diff, d = _distance(xyz, sh)
    inv = torch.where(d == 0, torch.zeros_like(d), 1.0 / d)                          # geom.py:331-378 divide_no_nan
    rij = diff * inv
    # CosCutOffEnvelope (geom.py:829-837; cutoff=None -> 1e8) - used only when the conv has a cutoff
    c = float(abs(cutoff)) if cutoff is not None else 1e8
    env = (torch.cos(torch.clamp(d, -c, c) * math.pi / c) + 1.0) * 0.5
    # BesselBasisLayer (geom.py:772-785)
    freq = p["bessel/frequencies"] if "bessel/frequencies" in p else \
        (math.pi * torch.arange(1, int(ba["num_radial"]) + 1, dtype=torch.float32)).to(dt)
    inv_cutoff = torch.tensor(np.float32(1.0 / float(ba["cutoff"]))).to(dt)
    ds = d * inv_cutoff
    pe = int(ba["envelope_exponent"]) + 1
    a_, b_, c_ = -(pe + 1) * (pe + 2) / 2, pe * (pe + 2), -pe * (pe + 1) / 2
    envelope = 1.0 / ds + a_ * ds ** (pe - 1) + b_ * ds ** pe + c_ * ds ** (pe + 1)
    rbf = torch.where(ds < 1, envelope, torch.zeros_like(ds)) * torch.sin(freq * ds)
    for i in range(depth):
        pre = "conv%d/" % i                                                           # painn_conv.py:97-115
        s = _dense(_dense(z, p, pre + "dense1", "swish"), p, pre + "phi")
        w = _dense(rbf, p, pre + "w")
        if cutoff is not None:
            w = w * env
        sw = s.index_select(0, sh[:, 1]) * w
        sw1, sw2, sw3 = torch.chunk(sw, 3, dim=-1)
        dz = _segment_sum(sw1, sh[:, 0], n_rows)
        vj = v.index_select(0, sh[:, 1])
        dv = _segment_sum(sw2.unsqueeze(-2) * vj + sw3.unsqueeze(-2) * rij.unsqueeze(-1), sh[:, 0], n_rows)
        z, v = z + dz, v + dv
        pre = "update%d/" % i                                                         # painn_conv.py:201-214
        v_v = v @ p[pre + "lin_v/kernel"]
        v_u = v @ p[pre + "lin_u/kernel"]
        v_prod = (v_u * v_v).sum(dim=1)
        v_norm = torch.sqrt(torch.relu(v_v.square().sum(dim=1)))
        a = _dense(_dense(torch.cat([z, v_norm], dim=-1), p, pre + "dense1", "swish"), p, pre + "a")
        a_vv, a_sv, a_ss = torch.chunk(a, 3, dim=-1)
        z, v = z + (v_prod * a_sv + a_ss), v + a_vv.unsqueeze(-2) * v_u
    out = _pool_nodes(z, ns)
    for k, act in enumerate(output_mlp_act):
        out = _dense(out, p, "output_mlp/%d" % k, act)
    return out


def energy_force(energy_fn, xyz, dtype=torch.float64, esp=None, desp_dr=None, is_physical_force=True):
    """``EnergyForceModel.call``, kgcnn/model/force.py:136-201, on flat ragged values.

    ``energy_fn(xyz_tensor[, esp_tensor]) -> (G, states)``.  Returns ``energy (G, states)`` and ``force (N, 3, states)``
    as NumPy arrays: ``de_dr = batch_jacobian(E, x)`` (:176-177), ``+ batch_jacobian(E, esp)[..., None] * desp_dr`` when the
    QM/MM inputs are given (:179-183), negated for ``is_physical_force`` (:185-186).  Squeezing the state axis
    (``output_squeeze_states``, :187-188) is left to the caller."""
    x = torch.from_numpy(np.ascontiguousarray(xyz)).to(dtype).requires_grad_(True)
    leaves = [x]
    if esp is not None:
        e_in = torch.from_numpy(np.ascontiguousarray(esp)).to(dtype).requires_grad_(True)
        leaves.append(e_in)
        eng = energy_fn(x, e_in)
    else:
        eng = energy_fn(x)
    states = int(eng.shape[1])
    cols = []
    for s in range(states):
        grads = torch.autograd.grad(eng[:, s].sum(), leaves, retain_graph=s + 1 < states, allow_unused=True)
        de_dr = grads[0] if grads[0] is not None else torch.zeros_like(x)
        if esp is not None and desp_dr is not None:
            de_desp = grads[1] if grads[1] is not None else torch.zeros_like(e_in)
            de_dr = de_dr + de_desp.unsqueeze(-1) * torch.from_numpy(np.ascontiguousarray(desp_dr)).to(dtype)
        cols.append(de_dr)
    de_dr = torch.stack(cols, dim=-1)
    if is_physical_force:
        de_dr = -de_dr
    return eng.detach().numpy(), de_dr.detach().numpy()


def schnet_energy_force(params, batch, dtype=torch.float64, is_physical_force=True, **kw):
    """Energy ``(G, states)`` and forces ``(N, 3)`` (single state squeezed) of a SchNet energy model on a synth batch."""
    p = to_torch(params, dtype)
    fn = lambda x: schnet_energy(p, batch["node_number"], x, batch["edge_indices"], batch["node_splits"],
                                 batch["edge_splits"], **kw)
    e, f = energy_force(fn, batch["node_coordinates"], dtype, is_physical_force=is_physical_force)
    return e, f[..., 0] if f.shape[-1] == 1 else f


def painn_energy_force(params, batch, dtype=torch.float64, is_physical_force=True, **kw):
    p = to_torch(params, dtype)
    fn = lambda x: painn_energy(p, batch["node_number"], x, batch["edge_indices"], batch["node_splits"],
                                batch["edge_splits"], **kw)
    e, f = energy_force(fn, batch["node_coordinates"], dtype, is_physical_force=is_physical_force)
    return e, f[..., 0] if f.shape[-1] == 1 else f
