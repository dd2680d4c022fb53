"""Synthetic ragged batches shaped like the reference's benchmark datasets (host side, NumPy only).

The GPU box has neither the reference tree nor any dataset, so the inputs of every BASELINE.json
config are generated here from ``numpy.random.default_rng(seed)``.  The
edge rule restates ``kgcnn.graph.adj.define_adjacency_from_distance`` (kgcnn/graph/adj.py:537-593,
as called by ``SetRange``, kgcnn/graph/preprocessor.py:288-314): connect ``i -> j`` when
``dist < max_distance`` AND ``j`` is among the ``max_neighbours + 1`` nearest entries of row ``i``
(exclusive mode), no self loops, indices in row-major ``(i, j)`` order - hence sorted by receiver.
tests/test_synth.py checks it against edge lists produced in the build container by the reference's
own function (tests/golden/radius_graph_cases.npz).
"""
import numpy as np


def distance_matrix(xyz):
    """kgcnn/graph/adj.py:466-483: ``sqrt(sum((b - a)^2))`` in the coordinate dtype."""
    xyz = np.asarray(xyz)
    c = xyz[:, None, :] - xyz[None, :, :]
    return np.sqrt(np.sum(np.square(c), axis=-1))


def radius_graph(xyz, max_distance=4.0, max_neighbours=30):
    """Edge list ``(m, 2)`` int64 of one molecule by the reference rule (exclusive, no self loops)."""
    dist = distance_matrix(xyz)
    n = dist.shape[-1]
    adj = np.ones_like(dist, dtype=bool)
    if max_distance is not None:
        adj &= dist < max_distance
    if max_neighbours is not None:
        k = min(int(max_neighbours), n)
        order = np.argsort(dist, axis=-1)[..., :k + 1]
        temp = np.zeros_like(dist, dtype=bool)
        np.put_along_axis(temp, order, True, axis=-1)
        adj &= temp
    adj[np.arange(n), np.arange(n)] = False
    ii, jj = np.nonzero(adj)  # row-major order == graph_indices[graph_adjacency]
    return np.stack([ii, jj], axis=-1).astype(np.int64)


def _splits(lengths):
    return np.concatenate([np.zeros(1, np.int64), np.cumsum(np.asarray(lengths, dtype=np.int64))])


def qm9_like_nodes(num_graphs=128, seed=1234, sigma=1.6):
    """Node side of :func:`qm9_like_batch` alone (same draws): numbers, coordinates and node row_splits.  The edge lists
    of large batches (BASELINE config 4: 100 000 molecules) are then built on the GPU by the engine's ``SetRange``
    (gcnn_keras_amd/graph/preprocessor.py), which applies the same rule as :func:`radius_graph`."""
    rng = np.random.default_rng(seed)
    rng_z = np.random.default_rng(seed + 1)
    z_vals = np.array([1, 6, 7, 8, 9], dtype=np.float32)
    z_p = np.array([.51, .35, .06, .07, .01])
    zs, xs, n_len = [], [], []
    for _ in range(num_graphs):
        n = int(np.clip(np.rint(rng.normal(18.0, 4.5)), 3, 29))
        xs.append(rng.normal(0.0, sigma, size=(n, 3)).astype(np.float32))
        zs.append(rng_z.choice(z_vals, size=n, p=z_p).astype(np.float32))
        n_len.append(n)
    return {"node_number": np.concatenate(zs), "node_coordinates": np.concatenate(xs, axis=0),
            "node_splits": _splits(n_len)}


def qm9_like_batch(num_graphs=128, seed=1234, sigma=1.6, max_distance=4.0, max_neighbours=30):
    """BASELINE config 2: QM9-shaped molecules.

    ``n_g = clip(round(N(18, 4.5^2)), 3, 29)``; ``Z`` from {1,6,7,8,9} w.p. {.51,.35,.06,.07,.01}
    as float32 (kgcnn/literature/Schnet.py:26 declares float node numbers); ``xyz ~ N(0, sigma^2 I)``;
    draw order per graph: n_g, then xyz (as in BASELINE.md's calibration: seed 1234 -> N=2301, M=26190);
    Z comes from a second stream (seed + 1).  Returns a dict of flat values + int64 row_splits.
    """
    out = qm9_like_nodes(num_graphs, seed, sigma)
    ns = out["node_splits"]
    es = [radius_graph(out["node_coordinates"][ns[g]:ns[g + 1]], max_distance=max_distance,
                       max_neighbours=max_neighbours) for g in range(num_graphs)]
    out["edge_indices"] = np.concatenate(es, axis=0).reshape(-1, 2).astype(np.int64)
    out["edge_splits"] = _splits([len(e) for e in es])
    return out


ASPIRIN_Z = np.array([6] * 9 + [1] * 8 + [8] * 4, dtype=np.float32)  # C9 H8 O4


def md17_like_batch(num_graphs=64, seed=2345, sigma=1.7, max_distance=5.0, max_neighbours=10000):
    """BASELINE config 3: 21-atom aspirin-composition molecules, cutoff 5 A, unlimited neighbours
    (training/hyper/hyper_md17.py:149)."""
    rng = np.random.default_rng(seed)
    zs, xs, es, n_len, e_len = [], [], [], [], []
    for _ in range(num_graphs):
        xyz = rng.normal(0.0, sigma, size=(21, 3)).astype(np.float32)
        ei = radius_graph(xyz, max_distance=max_distance, max_neighbours=max_neighbours)
        zs.append(ASPIRIN_Z.copy()); xs.append(xyz); es.append(ei); n_len.append(21); e_len.append(len(ei))
    return {
        "node_number": np.concatenate(zs), "node_coordinates": np.concatenate(xs, axis=0),
        "edge_indices": np.concatenate(es, axis=0).reshape(-1, 2).astype(np.int64),
        "node_splits": _splits(n_len), "edge_splits": _splits(e_len),
    }


def rescale_edge_weights_degree_sym(edge_indices, edge_weights):
    """``d_ii^-0.5 e_ij d_jj^-0.5`` with degree = occurrence count of the index in column 0 / 1
    (restates kgcnn/graph/adj.py:51-78)."""
    if len(edge_indices) == 0:
        return np.array([])
    row_val, row_cnt = np.unique(edge_indices[:, 0], return_counts=True)
    col_val, col_cnt = np.unique(edge_indices[:, 1], return_counts=True)
    d_row = np.zeros(len(edge_weights), dtype=edge_weights.dtype)
    d_col = np.zeros(len(edge_weights), dtype=edge_weights.dtype)
    d_row[row_val] = row_cnt
    d_col[col_val] = col_cnt
    with np.errstate(divide="ignore", invalid="ignore"):
        d_ii = np.nan_to_num(np.power(d_row, -0.5).flatten(), nan=0.0, posinf=0.0, neginf=0.0)
        d_jj = np.nan_to_num(np.power(d_col, -0.5).flatten(), nan=0.0, posinf=0.0, neginf=0.0)
    return d_ii[edge_indices[:, 0]][:, None] * edge_weights * d_jj[edge_indices[:, 1]][:, None]


def cora_like_graph(num_nodes=2708, attach=2, num_features=1433, density=0.0127, seed=4567, drop_pairs=134):
    """BASELINE config 5: one Cora-shaped graph.

    Barabasi-Albert preferential attachment (own implementation, ``attach`` edges per new node),
    ``drop_pairs`` undirected pairs removed uniformly -> both directions, sorted by (i, j), plus self
    loops, symmetric degree normalisation of unit weights (pipeline of training/hyper/hyper_cora.py:51-53),
    Bernoulli(density) features.
    """
    rng = np.random.default_rng(seed)
    targets = list(range(attach))
    repeated = []
    pairs = set()
    for src in range(attach, num_nodes):
        for t in set(targets):
            pairs.add((min(src, t), max(src, t)))
        repeated.extend(targets)
        repeated.extend([src] * attach)
        targets = []
        while len(targets) < attach:
            c = repeated[int(rng.integers(len(repeated)))]
            if c not in targets:
                targets.append(c)
    pairs = np.array(sorted(pairs), dtype=np.int64)
    if drop_pairs > 0:
        keep = np.ones(len(pairs), dtype=bool)
        keep[rng.choice(len(pairs), size=drop_pairs, replace=False)] = False
        pairs = pairs[keep]
    directed = np.concatenate([pairs, pairs[:, ::-1]], axis=0)
    loops = np.stack([np.arange(num_nodes), np.arange(num_nodes)], axis=-1)
    ei = np.concatenate([directed, loops], axis=0)
    order = np.lexsort((ei[:, 1], ei[:, 0]))
    ei = ei[order].astype(np.int64)
    w = rescale_edge_weights_degree_sym(ei, np.ones((len(ei), 1), dtype=np.float32)).astype(np.float32)
    x = (rng.random((num_nodes, num_features)) < density).astype(np.float32)
    return {
        "node_attributes": x, "edge_weights": w, "edge_indices": ei,
        "node_splits": _splits([num_nodes]), "edge_splits": _splits([len(ei)]),
    }


def toy_batch():
    """BASELINE config 1: the 3-graph README batch (README.md:84), F=3, values arange(18)/8."""
    idx = [[[0, 1], [1, 0]], [[0, 1], [1, 2], [2, 0]], [[0, 0]]]
    ei = np.concatenate([np.asarray(x, dtype=np.int64) for x in idx], axis=0)
    return {
        "node_attributes": (np.arange(18, dtype=np.float32).reshape(6, 3) / np.float32(8)),
        "edge_indices": ei,
        "node_splits": _splits([2, 3, 1]), "edge_splits": _splits([2, 3, 1]),
    }


def glorot_uniform(rng, fan_in, fan_out, shape=None):
    """Keras ``glorot_uniform``: U(-l, l), l = sqrt(6 / (fan_in + fan_out))."""
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape or (fan_in, fan_out)).astype(np.float32)


def schnet_params(seed=7, depth=3, units=128, emb_in=95, emb_out=64, bins=20, last_units=(128, 64),
                  out_units=(64, 1), random_bias=False):
    """Random-init SchNet weights in constructor order (shapes: SURVEY.md section 8a 'Parameter shapes';
    kgcnn/literature/Schnet.py:24-43, kgcnn/layers/conv/schnet_conv.py:50-51,136-139).
    Keras zero-initialises biases; ``random_bias=True`` draws small biases so parity tests exercise them."""
    rng = np.random.default_rng(seed)

    def bias(n):
        return (rng.uniform(-0.1, 0.1, size=n).astype(np.float32) if random_bias else np.zeros(n, np.float32))

    p = {"embedding": rng.uniform(-0.05, 0.05, size=(emb_in, emb_out)).astype(np.float32),
         "dense0/kernel": glorot_uniform(rng, emb_out, units), "dense0/bias": bias(units)}
    for i in range(depth):
        pre = "interaction%d/" % i
        p[pre + "cfconv/dense1/kernel"] = glorot_uniform(rng, bins, units)
        p[pre + "cfconv/dense1/bias"] = bias(units)
        p[pre + "cfconv/dense2/kernel"] = glorot_uniform(rng, units, units)
        p[pre + "cfconv/dense2/bias"] = bias(units)
        p[pre + "dense1/kernel"] = glorot_uniform(rng, units, units)
        p[pre + "dense2/kernel"] = glorot_uniform(rng, units, units)
        p[pre + "dense2/bias"] = bias(units)
        p[pre + "dense3/kernel"] = glorot_uniform(rng, units, units)
        p[pre + "dense3/bias"] = bias(units)
    fan = units
    for k, u in enumerate(last_units):
        p["last_mlp/%d/kernel" % k] = glorot_uniform(rng, fan, u)
        p["last_mlp/%d/bias" % k] = bias(u)
        fan = u
    for k, u in enumerate(out_units):
        p["output_mlp/%d/kernel" % k] = glorot_uniform(rng, fan, u)
        p["output_mlp/%d/bias" % k] = bias(u)
        fan = u
    return p


def painn_params(seed=8, depth=3, units=128, emb_in=95, num_radial=20, out_units=(128, 1), random_bias=False):
    """Random-init PaiNN weights (kgcnn/layers/conv/painn_conv.py:54-56,167-170; kgcnn/literature/PAiNN.py:24-42)."""
    rng = np.random.default_rng(seed)

    def bias(n):
        return (rng.uniform(-0.1, 0.1, size=n).astype(np.float32) if random_bias else np.zeros(n, np.float32))

    p = {"embedding": rng.uniform(-0.05, 0.05, size=(emb_in, units)).astype(np.float32),
         "bessel/frequencies": (np.pi * np.arange(1, num_radial + 1, dtype=np.float32))}
    for i in range(depth):
        c = "conv%d/" % i
        p[c + "dense1/kernel"] = glorot_uniform(rng, units, units); p[c + "dense1/bias"] = bias(units)
        p[c + "phi/kernel"] = glorot_uniform(rng, units, 3 * units); p[c + "phi/bias"] = bias(3 * units)
        p[c + "w/kernel"] = glorot_uniform(rng, num_radial, 3 * units); p[c + "w/bias"] = bias(3 * units)
        u = "update%d/" % i
        p[u + "dense1/kernel"] = glorot_uniform(rng, 2 * units, units); p[u + "dense1/bias"] = bias(units)
        p[u + "lin_u/kernel"] = glorot_uniform(rng, units, units)
        p[u + "lin_v/kernel"] = glorot_uniform(rng, units, units)
        p[u + "a/kernel"] = glorot_uniform(rng, units, 3 * units); p[u + "a/bias"] = bias(3 * units)
    fan = units
    for k, uu in enumerate(out_units):
        p["output_mlp/%d/kernel" % k] = glorot_uniform(rng, fan, uu)
        p["output_mlp/%d/bias" % k] = bias(uu)
        fan = uu
    return p


def gcn_params(seed=9, depth=3, in_features=1433, units=64, out_units=(64, 32, 7), random_bias=False):
    """Random-init GCN weights (kgcnn/literature/GCN.py:95-97, kgcnn/layers/conv/gcn_conv.py:63-66)."""
    rng = np.random.default_rng(seed)

    def bias(n):
        return (rng.uniform(-0.1, 0.1, size=n).astype(np.float32) if random_bias else np.zeros(n, np.float32))

    p = {"dense0/kernel": glorot_uniform(rng, in_features, units), "dense0/bias": bias(units)}
    for i in range(depth):
        p["gcn%d/kernel" % i] = glorot_uniform(rng, units, units)
        p["gcn%d/bias" % i] = bias(units)
    fan = units
    for k, u in enumerate(out_units):
        p["output_mlp/%d/kernel" % k] = glorot_uniform(rng, fan, u)
        p["output_mlp/%d/bias" % k] = bias(u)
        fan = u
    return p
