"""CPU oracle: a NumPy restatement of the kgcnn ragged scatter-gather hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The product path (``gcnn_keras_amd``) never imports anything from ``oracle/``.

Every function restates, op for op and in float32 unless the inputs are float64, the TF
op sequence of one reference function and cites it as ``file:line`` relative to the
reference tree (Tacitus523/gcnn_keras, kgcnn 2.2.3).

Pinning status.  TensorFlow is not installed, so the reference
path cannot be executed here; the arithmetic of the path lives in the third-party
dependency ``tensorflow`` (pins: ``>=2.9.0`` setup.py:24, ``==2.12.0`` env_linux.yml:228).
The oracle is pinned by the reference's own known answers (tests/test_oracle_pins.py):

  1. kgcnn/ops/partition.py:112-120   docstring example of ``partition_row_indexing``
  2. test/test_gather.py:11-44        ``GatherNodes`` on a 2-graph batch vs NumPy indexing
  3. test/test_conv_attention.py:34-43 attention pooling known answer ``100/(e+1)``
  4. test/test_geom.py:79-128 + test/assets/bessel_basis_reference.npz (copied as data to
     tests/golden/bessel_basis_reference.npz)  NodePosition -> NodeDistanceEuclidean ->
     BesselBasisLayer(10, 5.0)
  5. test/test_conv_dmpnn.py:11-28    ``DMPNNGatherEdgesPairs`` on its two-graph fixture

Everything else (PoolingLocalEdges sum/mean/max/min values, weighted pooling, PoolingNodes,
SchNetCFconv / SchNetInteraction, PAiNNconv / PAiNNUpdate, GCN, GIN, GAT heads, MEGNet block, GraphSAGE,
layer normalisation, Gaussian basis, whole-model outputs) is **parity unpinned** by the reference's own tests: those results are checked
against this restatement only, cross-checked by an independent torch-CPU formulation in
tests/test_oracle_crosscheck.py.

TF semantics relied upon (documented TF behaviour, recalled; TF source is not in the tree):
sorted ``tf.math.segment_*`` size the output by the last id + 1 and fill missing ids with 0;
``tf.argsort(stable=True)``; ``tf.scatter_nd`` sums duplicates; ``tf.nn.softplus`` uses the
thresholded ``log1p(exp(x))`` form; ``divide_no_nan`` returns 0 where the divisor is 0.
"""
from collections import namedtuple

import numpy as np

#: Minimal ragged carrier of ragged_rank 1: flat ``values`` plus int64 ``row_splits``.
R = namedtuple("R", ["values", "row_splits"])


def ragged_from_row_lengths(values, row_lengths):
    row_lengths = np.asarray(row_lengths, dtype=np.int64)
    splits = np.concatenate([np.zeros(1, dtype=np.int64), np.cumsum(row_lengths, dtype=np.int64)])
    return R(np.asarray(values), splits)


def ragged_from_list(rows, dtype, inner_shape=()):
    lens = [len(r) for r in rows]
    if sum(lens) == 0:
        vals = np.zeros((0,) + tuple(inner_shape), dtype=dtype)
    else:
        vals = np.concatenate([np.asarray(r, dtype=dtype).reshape((len(r),) + tuple(inner_shape)) for r in rows], axis=0)
    return ragged_from_row_lengths(vals, lens)


def row_lengths(r):
    return r.row_splits[1:] - r.row_splits[:-1]


def value_rowids(r):
    lens = row_lengths(r)
    return np.repeat(np.arange(len(lens), dtype=np.int64), lens)


def ragged_rows(r):
    """List of per-row value arrays (``ragged[i]``)."""
    return [r.values[r.row_splits[i]:r.row_splits[i + 1]] for i in range(len(r.row_splits) - 1)]


# ----------------------------------------------------------------------------------------
# kgcnn/ops/partition.py
# ----------------------------------------------------------------------------------------

def change_partition_by_name(in_partition, in_partition_type, out_partition_type):
    """kgcnn/ops/partition.py:5-93."""
    p = np.asarray(in_partition)
    lengths_n = ["row_length", "row_lengths"]
    splits_n = ["row_split", "row_splits"]
    starts_n = ["row_start", "row_starts"]
    limits_n = ["row_limit", "row_limits"]

    def _seg_count(ids):
        # tf.math.segment_sum(ones_like(ids), ids): rows = last id + 1
        if len(ids) == 0:
            return np.zeros(0, dtype=ids.dtype)
        return np.bincount(ids, minlength=int(ids[-1]) + 1).astype(ids.dtype)

    if in_partition_type == out_partition_type:
        return p
    if in_partition_type in lengths_n and out_partition_type in splits_n:
        return np.pad(np.cumsum(p, dtype=p.dtype), (1, 0))
    if in_partition_type in lengths_n and out_partition_type == "value_rowids":
        return np.repeat(np.arange(p.shape[0], dtype=np.int32), p)  # tf.range -> int32 (partition.py:29)
    if in_partition_type in lengths_n and out_partition_type in starts_n:
        return np.cumsum(p, dtype=p.dtype) - p
    if in_partition_type in lengths_n and out_partition_type in limits_n:
        return np.cumsum(p, dtype=p.dtype)
    if in_partition_type in splits_n and out_partition_type in lengths_n:
        return p[1:] - p[:-1]
    if in_partition_type in splits_n and out_partition_type == "value_rowids":
        part_sum = p[1:] - p[:-1]
        return np.repeat(np.arange(part_sum.shape[0], dtype=np.int32), part_sum)
    if in_partition_type in splits_n and out_partition_type in limits_n:
        return p[1:]
    if in_partition_type in splits_n and out_partition_type in starts_n:
        return p[:-1]
    if in_partition_type == "value_rowids" and out_partition_type in lengths_n:
        return _seg_count(p)
    if in_partition_type == "value_rowids" and out_partition_type in splits_n:
        return np.pad(np.cumsum(_seg_count(p), dtype=p.dtype), (1, 0))
    if in_partition_type == "value_rowids" and out_partition_type in limits_n:
        return np.cumsum(_seg_count(p), dtype=p.dtype)
    if in_partition_type == "value_rowids" and out_partition_type in starts_n:
        c = _seg_count(p)
        return np.cumsum(c, dtype=p.dtype) - c
    if in_partition_type in starts_n:
        raise ValueError("Can not infer partition scheme from row_starts alone, missing nvals")
    if in_partition_type in limits_n and out_partition_type in lengths_n:
        s = np.pad(p, (1, 0))
        return s[1:] - s[:-1]
    if in_partition_type in limits_n and out_partition_type == "value_rowids":
        s = np.pad(p, (1, 0))
        part_sum = s[1:] - s[:-1]
        return np.repeat(np.arange(part_sum.shape[0], dtype=np.int32), part_sum)
    if in_partition_type in limits_n and out_partition_type in splits_n:
        return np.pad(p, (1, 0))
    if in_partition_type in limits_n and out_partition_type in starts_n:
        return np.pad(p, (1, 0))[:-1]
    raise TypeError("Unknown partition scheme, use: 'value_rowids', 'row_splits', 'row_lengths', etc.")


def partition_row_indexing(tensor_index, part_target, part_index, partition_type_target, partition_type_index,
                           from_indexing="sample", to_indexing="batch"):
    """kgcnn/ops/partition.py:97-162: ``idx +/- node_row_splits[graph_of_edge]`` in the index dtype."""
    tensor_index = np.asarray(tensor_index)
    if to_indexing == from_indexing:
        return tensor_index
    nod_splits = change_partition_by_name(part_target, partition_type_target, "row_splits")
    edge_ids = change_partition_by_name(part_index, partition_type_index, "value_rowids")
    shift_index = nod_splits[edge_ids]
    for _ in range(1, tensor_index.ndim):
        shift_index = np.expand_dims(shift_index, axis=-1)
    if to_indexing == "batch" and from_indexing == "sample":
        return tensor_index + shift_index.astype(tensor_index.dtype)
    if to_indexing == "sample" and from_indexing == "batch":
        return tensor_index - shift_index.astype(tensor_index.dtype)
    raise TypeError("ERROR:kgcnn: Unknown index change, use: 'sample', 'batch', ...")


# ----------------------------------------------------------------------------------------
# kgcnn/ops/segment.py, kgcnn/ops/scatter.py
# ----------------------------------------------------------------------------------------

def _num_segments(ids):
    return int(ids[-1]) + 1 if len(ids) > 0 else 0


def segment_sum(data, ids):
    """``tf.math.segment_sum`` for sorted ids: rows = last id + 1, gaps 0, sequential accumulation
    in row order (``np.add.at`` is unbuffered and applies the updates in index order)."""
    data = np.asarray(data)
    out = np.zeros((_num_segments(ids),) + data.shape[1:], dtype=data.dtype)
    np.add.at(out, np.asarray(ids, dtype=np.int64), data)
    return out


def segment_mean(data, ids):
    data = np.asarray(data)
    n = _num_segments(ids)
    s = segment_sum(data, ids)
    cnt = np.bincount(np.asarray(ids, dtype=np.int64), minlength=n).astype(data.dtype)
    cnt = cnt.reshape((n,) + (1,) * (data.ndim - 1))
    with np.errstate(divide="ignore", invalid="ignore"):
        out = s / cnt
    return np.where(cnt > 0, out, np.zeros_like(out)).astype(data.dtype)


def _segment_extreme(data, ids, ufunc, init):
    data = np.asarray(data)
    n = _num_segments(ids)
    ids = np.asarray(ids, dtype=np.int64)
    out = np.full((n,) + data.shape[1:], init, dtype=data.dtype)
    ufunc.at(out, ids, data)
    present = np.bincount(ids, minlength=n) > 0
    out[~present] = 0
    return out


def segment_max(data, ids):
    return _segment_extreme(data, ids, np.maximum, -np.inf)


def segment_min(data, ids):
    return _segment_extreme(data, ids, np.minimum, np.inf)


def segment_ops_by_name(segment_name, data, segment_ids):
    """kgcnn/ops/segment.py:28-52."""
    if segment_name in ["segment_mean", "mean", "reduce_mean"]:
        return segment_mean(data, segment_ids)
    if segment_name in ["segment_sum", "sum", "reduce_sum"]:
        return segment_sum(data, segment_ids)
    if segment_name in ["segment_max", "max", "reduce_max"]:
        return segment_max(data, segment_ids)
    if segment_name in ["segment_min", "min", "reduce_min"]:
        return segment_min(data, segment_ids)
    raise TypeError("Unknown segment operation, choose: 'segment_mean', 'segment_sum', ...")


def segment_softmax(data, segment_ids, normalize=True):
    """kgcnn/ops/segment.py:5-24."""
    data = np.asarray(data)
    if normalize:
        data_segment_max = segment_max(data, segment_ids)
        data = data - data_segment_max[segment_ids]
    data_exp = np.exp(data)
    data_exp_segment_sum = segment_sum(data_exp, segment_ids)
    return data_exp / data_exp_segment_sum[segment_ids]


def tensor_scatter_nd_ops_by_name(segment_name, tensor, indices, updates):
    """kgcnn/ops/scatter.py:5-26 (``tf.tensor_scatter_nd_{add,max,min}``)."""
    out = np.array(tensor, copy=True)
    idx = tuple(np.asarray(indices)[:, k] for k in range(np.asarray(indices).shape[1]))
    if segment_name in ["segment_sum", "sum", "reduce_sum", "add"]:
        np.add.at(out, idx, updates)
    elif segment_name in ["segment_max", "max", "reduce_max"]:
        np.maximum.at(out, idx, updates)
    elif segment_name in ["segment_min", "min", "reduce_min"]:
        np.minimum.at(out, idx, updates)
    else:
        raise TypeError("Unknown pooling, choose: 'mean', 'sum', ...")
    return out


# ----------------------------------------------------------------------------------------
# Activations: kgcnn/ops/activ.py:6-15 and the Keras strings used by SchNet / PaiNN / GCN
# ----------------------------------------------------------------------------------------

def softplus(x):
    """``tf.nn.softplus``: x if x > -thr ; exp(x) if x < thr ; else log1p(exp(x)),
    thr = log(eps) + 2 (TF ``softplus_op.h`` functor, recalled)."""
    x = np.asarray(x)
    thr = np.log(np.finfo(x.dtype).eps).astype(x.dtype) + x.dtype.type(2)
    with np.errstate(over="ignore"):
        ex = np.exp(x)
        mid = np.log1p(ex)
    return np.where(x > -thr, x, np.where(x < thr, ex, mid)).astype(x.dtype)


def shifted_softplus(x):
    """kgcnn/ops/activ.py:15: ``softplus(x) - log(2.0)`` in the input dtype."""
    x = np.asarray(x)
    return (softplus(x) - np.log(x.dtype.type(2.0))).astype(x.dtype)


def sigmoid(x):
    x = np.asarray(x)
    with np.errstate(over="ignore"):
        return (x.dtype.type(1) / (x.dtype.type(1) + np.exp(-x))).astype(x.dtype)


def swish(x):
    x = np.asarray(x)
    return (x * sigmoid(x)).astype(x.dtype)


def softmax_last(x):
    x = np.asarray(x)
    e = np.exp(x - np.max(x, axis=-1, keepdims=True))
    return (e / np.sum(e, axis=-1, keepdims=True)).astype(x.dtype)


def leaky_relu(x, alpha=0.05):
    """kgcnn/ops/activ.py:59-80 ``kgcnn>leaky_relu`` (default alpha 0.05)."""
    x = np.asarray(x)
    return np.where(x >= 0, x, x.dtype.type(alpha) * x).astype(x.dtype)


def softplus2(x):
    """kgcnn/ops/activ.py:19-29: ``relu(x) + log(0.5 * exp(-|x|) + 0.5)``."""
    x = np.asarray(x)
    half = np.asarray(0.5, x.dtype)
    return (np.maximum(x, 0) + np.log(half * np.exp(-np.abs(x)) + half)).astype(x.dtype)


ACTIVATIONS = {
    None: lambda x: x,
    "linear": lambda x: x,
    "relu": lambda x: np.maximum(x, 0).astype(np.asarray(x).dtype),
    "kgcnn>shifted_softplus": shifted_softplus,
    "shifted_softplus": shifted_softplus,
    "softplus": softplus,
    "swish": swish,
    "sigmoid": sigmoid,
    "tanh": np.tanh,
    "softmax": softmax_last,
    "kgcnn>leaky_relu": leaky_relu,
    "kgcnn>softplus2": softplus2,
    "softplus2": softplus2,
    "selu": lambda x: (np.asarray(x).dtype.type(1.05070098) * np.where(
        np.asarray(x) > 0, x, np.asarray(x).dtype.type(1.67326324) * (np.exp(np.minimum(x, 0)) - 1))).astype(np.asarray(x).dtype),
}


def activation(name, x):
    if callable(name):
        return name(x)
    return ACTIVATIONS[name](x)


# ----------------------------------------------------------------------------------------
# kgcnn/layers/modules.py, kgcnn/layers/mlp.py
# ----------------------------------------------------------------------------------------

def dense_values(x, kernel, bias=None, act=None):
    """Keras ``Dense`` on the last axis of a values tensor (kgcnn/layers/modules.py:74-87):
    ``act(x @ kernel + bias)``, kernel layout ``(in, units)``."""
    x = np.asarray(x)
    y = np.matmul(x, kernel.astype(x.dtype))
    if bias is not None:
        y = y + bias.astype(x.dtype)
    return activation(act, y)


def dense(r, kernel, bias=None, act=None):
    if isinstance(r, R):
        return R(dense_values(r.values, kernel, bias, act), r.row_splits)
    return dense_values(r, kernel, bias, act)


def mlp(r, layers):
    """kgcnn/layers/mlp.py:299-316 with dropout / normalisation off: Dense(linear) then Activation.
    ``layers``: list of ``(kernel, bias_or_None, activation_name)``."""
    x = r
    for kernel, bias, act in layers:
        x = dense(x, kernel, bias, None)
        if isinstance(x, R):
            x = R(activation(act, x.values), x.row_splits)
        else:
            x = activation(act, x)
    return x


def embedding(r, table):
    """Keras ``Embedding`` on ragged float node numbers (kgcnn/layers/modules.py:526-528;
    kgcnn/literature/Schnet.py:26 declares float32): the input is cast to int32, then looked up."""
    idx = np.asarray(r.values).astype(np.int32)
    return R(table[idx], r.row_splits)


def lazy_add(rs):
    out = rs[0].values
    for x in rs[1:]:
        out = out + x.values
    return R(out, rs[0].row_splits)


def lazy_subtract(rs):
    return R(rs[0].values - rs[1].values, rs[0].row_splits)


def lazy_multiply(rs):
    out = rs[0].values
    for x in rs[1:]:
        out = out * x.values
    return R(out, rs[0].row_splits)


def lazy_concatenate(rs, axis=-1):
    """kgcnn/layers/modules.py:305-364 on values; ragged axis>1 maps to values axis-1 (base.py:131-144)."""
    nd = rs[0].values.ndim + 1
    ax = axis if axis >= 0 else axis + nd
    return R(np.concatenate([x.values for x in rs], axis=ax - 1), rs[0].row_splits)


def expand_dims(r, axis=-1):
    """kgcnn/layers/modules.py:368-416 (axis normalised against rank+1 of the ragged tensor)."""
    nd = r.values.ndim + 1
    ax = axis if axis >= 0 else axis + nd + 1
    return R(np.expand_dims(r.values, axis=ax - 1), r.row_splits)


# ----------------------------------------------------------------------------------------
# kgcnn/layers/gather.py
# ----------------------------------------------------------------------------------------

def _shift(nodes, idx):
    return partition_row_indexing(idx.values, nodes.row_splits, row_lengths(idx),
                                  partition_type_target="row_splits", partition_type_index="row_length",
                                  to_indexing="batch", from_indexing="sample")


def gather_nodes(nodes, idx, concat_axis=2, split_axis=None):
    """``GatherEmbedding`` fast path, kgcnn/layers/gather.py:69-99.  Default output ``[x_i || x_j]``."""
    if split_axis is not None and concat_axis is not None:
        raise ValueError("Can not both split and concatenate new index axis. At least one must be `None`.")
    disjoint = _shift(nodes, idx)
    out = nodes.values[disjoint]  # (M, K, F...)
    if concat_axis == 2:
        out = np.concatenate([out[:, i] for i in range(idx.values.shape[1])], axis=1)
        return R(out, idx.row_splits)
    if split_axis == 2:
        return [R(out[:, i], idx.row_splits) for i in range(idx.values.shape[1])]
    return R(out, idx.row_splits)


def gather_nodes_selection(nodes, idx, selection_index):
    """``GatherEmbeddingSelection``, kgcnn/layers/gather.py:217-231: always returns a list."""
    if isinstance(selection_index, int):
        selection_index = [selection_index]
    indexlist = _shift(nodes, idx)
    return [R(nodes.values[indexlist[:, i]], idx.row_splits) for i in selection_index]


def gather_nodes_ingoing(nodes, idx):
    """kgcnn/layers/gather.py:249-282 (selection_index 0 = receiver i)."""
    return gather_nodes_selection(nodes, idx, 0)[0]


def gather_nodes_outgoing(nodes, idx):
    """kgcnn/layers/gather.py:286-319 (selection_index 1 = sender j)."""
    return gather_nodes_selection(nodes, idx, 1)[0]


def gather_state(state, target):
    """kgcnn/layers/gather.py:323-375: ``tf.repeat(state, target.row_lengths(), axis=0)``."""
    lens = row_lengths(target)
    return R(np.repeat(np.asarray(state), lens, axis=0), target.row_splits)


# ----------------------------------------------------------------------------------------
# kgcnn/layers/pooling.py
# ----------------------------------------------------------------------------------------

def _scatter_pad(out, n_rows):
    """``tf.scatter_nd(range(rows)[:, None], out, (N, ...))`` (pooling.py:74-76): zero-pad tail rows."""
    padded = np.zeros((n_rows,) + out.shape[1:], dtype=out.dtype)
    padded[:out.shape[0]] = out
    return padded


def pooling_local_edges(nodes, edges, idx, pooling_method="mean", pooling_index=0,
                        is_sorted=False, has_unconnected=True):
    """``PoolingLocalEdges.call``, kgcnn/layers/pooling.py:37-79."""
    shiftind = _shift(nodes, idx)
    nodind = shiftind[:, pooling_index]
    dens = edges.values
    if not is_sorted:
        node_order = np.argsort(nodind, axis=0, kind="stable")
        nodind = nodind[node_order]
        dens = dens[node_order]
    out = segment_ops_by_name(pooling_method, dens, nodind)
    if has_unconnected:
        out = _scatter_pad(out, nodes.values.shape[0])
    return R(out, nodes.row_splits)


def pooling_weighted_local_edges(nodes, edges, idx, weights, pooling_method="mean", normalize_by_weights=False,
                                 pooling_index=0, is_sorted=False, has_unconnected=True):
    """``PoolingWeightedLocalEdges.call``, kgcnn/layers/pooling.py:126-176."""
    shiftind = _shift(nodes, idx)
    wval = weights.values
    dens = edges.values * wval
    nodind = shiftind[:, pooling_index]
    if not is_sorted:
        node_order = np.argsort(nodind, axis=0, kind="stable")
        nodind = nodind[node_order]
        dens = dens[node_order]
        wval = wval[node_order]
    get = segment_ops_by_name(pooling_method, dens, nodind)
    if normalize_by_weights:
        den = segment_sum(wval, nodind)
        with np.errstate(divide="ignore", invalid="ignore"):
            q = get / den
        get = np.where(den == 0, np.zeros_like(q), q).astype(get.dtype)
    if has_unconnected:
        get = _scatter_pad(get, nodes.values.shape[0])
    return R(get, nodes.row_splits)


def pooling_nodes(nodes, pooling_method="mean"):
    """``PoolingEmbedding.call``, kgcnn/layers/pooling.py:203-219: dense ``(max(rowid)+1, F)``."""
    return segment_ops_by_name(pooling_method, nodes.values, value_rowids(nodes))


def pooling_weighted_nodes(nodes, weights, pooling_method="mean"):
    """``PoolingWeightedEmbedding.call``, kgcnn/layers/pooling.py:257-276."""
    return segment_ops_by_name(pooling_method, nodes.values * weights.values, value_rowids(nodes))


def pooling_local_edges_attention(nodes, edges, attention, idx, pooling_index=0, is_sorted=False,
                                  has_unconnected=True):
    """``PoolingLocalEdgesAttention.call``, kgcnn/layers/pooling.py:494-541."""
    shiftind = _shift(nodes, idx)
    nodind = shiftind[:, pooling_index]
    dens = edges.values
    ats = attention.values
    if not is_sorted:
        node_order = np.argsort(nodind, axis=0, kind="stable")
        nodind = nodind[node_order]
        dens = dens[node_order]
        ats = ats[node_order]
    ats = segment_softmax(ats, nodind)
    get = segment_sum(dens * ats, nodind)
    if has_unconnected:
        get = _scatter_pad(get, nodes.values.shape[0])
    return R(get, nodes.row_splits)


def pooling_nodes_attention(nodes, attention):
    """``PoolingEmbeddingAttention.call``, kgcnn/layers/pooling.py:570-591."""
    batchi = value_rowids(nodes)
    ats = segment_softmax(attention.values, batchi)
    return segment_sum(nodes.values * ats, batchi)


def relational_pooling_local_edges(nodes, edges, idx, edge_relation, num_relations, pooling_method="sum",
                                   pooling_index=0):
    """``RelationalPoolingLocalEdges.call``, kgcnn/layers/pooling.py:630-668."""
    shiftind = _shift(nodes, idx)
    indices = shiftind[:, pooling_index]
    relations = edge_relation.values.astype(indices.dtype)
    scatter_indices = np.stack([indices, relations], axis=-1)
    out_tensor = np.zeros((nodes.values.shape[0], num_relations) + edges.values.shape[1:], dtype=edges.values.dtype)
    out = tensor_scatter_nd_ops_by_name(pooling_method, out_tensor, scatter_indices, edges.values)
    return R(out, nodes.row_splits)


# ----------------------------------------------------------------------------------------
# kgcnn/layers/geom.py
# ----------------------------------------------------------------------------------------

def node_position(xyz, idx, selection_index=(0, 1)):
    """``NodePosition``, kgcnn/layers/geom.py:14-73 = ``GatherNodesSelection([0, 1])``."""
    return gather_nodes_selection(xyz, idx, list(selection_index))


def euclidean_norm_values(x, axis=-1, keepdims=False, invert_norm=False, add_eps=False, no_nan=True,
                          square_norm=False, epsilon=1e-7):
    """``EuclideanNorm._compute_euclidean_norm``, kgcnn/layers/geom.py:166-193: ``sqrt(relu(sum(x^2)))``.
    ``epsilon`` = ``ks.backend.epsilon()`` default 1e-7."""
    x = np.asarray(x)
    out = np.maximum(np.sum(np.square(x), axis=axis, keepdims=keepdims), 0).astype(x.dtype)
    if add_eps:
        out = out + x.dtype.type(epsilon)
    if not square_norm:
        out = np.sqrt(out)
    if invert_norm:
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = x.dtype.type(1) / out
        if no_nan:
            inv = np.where(out == 0, np.zeros_like(inv), inv)
        out = inv.astype(x.dtype)
    return out


def euclidean_norm(r, axis=-1, **kw):
    """Ragged wrapper: ``axis`` refers to the ragged tensor (batch axis 0), mapped to values axis-1."""
    nd = r.values.ndim + 1
    ax = axis if axis >= 0 else axis + nd
    return R(euclidean_norm_values(r.values, axis=ax - 1, **kw), r.row_splits)


def scalar_product(a, b, axis=-1):
    """``ScalarProduct``, kgcnn/layers/geom.py:250-261."""
    nd = a.values.ndim + 1
    ax = axis if axis >= 0 else axis + nd
    return R(np.sum(a.values * b.values, axis=ax - 1), a.row_splits)


def node_distance_euclidean(pos1, pos2, add_eps=False, no_nan=True):
    """``NodeDistanceEuclidean``, kgcnn/layers/geom.py:285-327: ``||x_1 - x_2||`` with keepdims."""
    diff = lazy_subtract([pos1, pos2])
    return euclidean_norm(diff, axis=2, keepdims=True, add_eps=add_eps, no_nan=no_nan)


def edge_direction_normalized(pos1, pos2, add_eps=False, no_nan=True):
    """``EdgeDirectionNormalized``, kgcnn/layers/geom.py:331-378: ``d * divide_no_nan(1, ||d||)``."""
    diff = lazy_subtract([pos1, pos2])
    norm = euclidean_norm(diff, axis=2, keepdims=True, invert_norm=True, add_eps=add_eps, no_nan=no_nan)
    return lazy_multiply([diff, norm])


def gauss_basis(d, bins=20, distance=4.0, sigma=0.4, offset=0.0):
    """``GaussBasisLayer._compute_gauss_basis``, kgcnn/layers/geom.py:554-571."""
    x = np.asarray(d.values)
    dt = x.dtype
    gamma = 1 / sigma / sigma / 2  # geom.py:549 (python float)
    gbs = np.arange(0, int(bins), 1, dtype=dt) / dt.type(float(bins)) * dt.type(distance)
    out = x - dt.type(offset)
    out = np.square(out - gbs) * dt.type(gamma * (-1.0))
    return R(np.exp(out).astype(dt), d.row_splits)


def bessel_basis(d, num_radial, cutoff, envelope_exponent=5, frequencies=None):
    """``BesselBasisLayer.expand_bessel_basis`` + ``envelope``, kgcnn/layers/geom.py:772-785.
    ``frequencies`` default ``pi * arange(1, num_radial + 1)`` float32 (geom.py:766-770)."""
    x = np.asarray(d.values)
    dt = x.dtype
    if frequencies is None:
        frequencies = (np.pi * np.arange(1, num_radial + 1, dtype=np.float32))
    frequencies = np.asarray(frequencies).astype(dt)
    inv_cutoff = dt.type(np.float32(1 / cutoff))
    d_scaled = x * inv_cutoff
    p = envelope_exponent + 1
    a = -(p + 1) * (p + 2) / 2
    b = p * (p + 2)
    c = -p * (p + 1) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        env_val = (dt.type(1.0) / d_scaled + dt.type(a) * d_scaled ** (p - 1) + dt.type(b) * d_scaled ** p
                   + dt.type(c) * d_scaled ** (p + 1))
    d_cutoff = np.where(d_scaled < 1, env_val, np.zeros_like(d_scaled))
    out = d_cutoff * np.sin(frequencies * d_scaled)
    return R(out.astype(dt), d.row_splits)


def cos_cutoff_envelope(d, cutoff):
    """``CosCutOffEnvelope``, kgcnn/layers/geom.py:829-837 (``cutoff=None`` -> 1e8)."""
    x = np.asarray(d.values)
    dt = x.dtype
    cutoff = float(np.abs(cutoff)) if cutoff is not None else 1e8
    fc = np.clip(x, dt.type(-cutoff), dt.type(cutoff))
    fc = (np.cos(fc * dt.type(np.pi) / dt.type(cutoff)) + dt.type(1)) * dt.type(0.5)
    return R(fc.astype(dt), d.row_splits)


# ----------------------------------------------------------------------------------------
# kgcnn/layers/conv/schnet_conv.py, kgcnn/literature/Schnet.py
# ----------------------------------------------------------------------------------------

def schnet_cfconv(node, edge, idx, p, act="kgcnn>shifted_softplus", cfconv_pool="sum"):
    """``SchNetCFconv.call``, kgcnn/layers/conv/schnet_conv.py:73-79.
    ``p``: dict with dense1/dense2 kernel + bias."""
    x = dense(edge, p["dense1/kernel"], p.get("dense1/bias"), act)
    x = dense(x, p["dense2/kernel"], p.get("dense2/bias"), "linear")
    node2exp = gather_nodes_outgoing(node, idx)
    x = lazy_multiply([node2exp, x])
    return pooling_local_edges(node, x, idx, pooling_method=cfconv_pool)


def schnet_interaction(node, edge, idx, p, act="kgcnn>shifted_softplus", cfconv_pool="sum"):
    """``SchNetInteraction.call``, kgcnn/layers/conv/schnet_conv.py:159-165."""
    x = dense(node, p["dense1/kernel"], None, "linear")
    x = schnet_cfconv(x, edge, idx, {k[len("cfconv/"):]: v for k, v in p.items() if k.startswith("cfconv/")},
                      act=act, cfconv_pool=cfconv_pool)
    x = dense(x, p["dense2/kernel"], p.get("dense2/bias"), act)
    x = dense(x, p["dense3/kernel"], p.get("dense3/bias"), "linear")
    return lazy_add([node, x])


def _sub(params, prefix):
    return {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}


def schnet_forward(params, node_number, xyz, idx, depth=3, gauss_args=None, act="kgcnn>shifted_softplus",
                   cfconv_pool="sum", node_pooling="sum",
                   last_mlp_act=("kgcnn>shifted_softplus", "kgcnn>shifted_softplus"),
                   output_mlp_act=("kgcnn>shifted_softplus", "linear"), return_intermediate=False):
    """``kgcnn.literature.Schnet.make_model`` forward, kgcnn/literature/Schnet.py:104-148
    (graph output, ``make_distance=True, expand_distance=True``)."""
    gauss_args = gauss_args or {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4}
    inter = {}
    # OptionalInputEmbedding (kgcnn/layers/modules.py:526-534): 2-D node attributes pass through unchanged
    n = embedding(node_number, params["embedding"]) if "embedding" in params else node_number
    pos1, pos2 = node_position(xyz, idx)
    ed = node_distance_euclidean(pos1, pos2)
    inter["distance"] = ed.values
    ed = gauss_basis(ed, **gauss_args)
    inter["rbf"] = ed.values
    n = dense(n, params["dense0/kernel"], params["dense0/bias"], "linear")
    for i in range(depth):
        n = schnet_interaction(n, ed, idx, _sub(params, "interaction%d/" % i), act=act, cfconv_pool=cfconv_pool)
        inter["n%d" % i] = n.values
    n = mlp(n, [(params["last_mlp/%d/kernel" % k], params.get("last_mlp/%d/bias" % k), last_mlp_act[k])
                for k in range(len(last_mlp_act))])
    inter["last_mlp"] = n.values
    out = pooling_nodes(n, node_pooling)
    inter["pooled"] = out
    out = mlp(out, [(params["output_mlp/%d/kernel" % k], params.get("output_mlp/%d/bias" % k), output_mlp_act[k])
                    for k in range(len(output_mlp_act))])
    if return_intermediate:
        return out, inter
    return out


# ----------------------------------------------------------------------------------------
# kgcnn/layers/conv/painn_conv.py, kgcnn/literature/PAiNN.py
# ----------------------------------------------------------------------------------------

def equivariant_initialize(z, dim=3, method="zeros", value=1.0, epsilon=1e-7):
    """``EquivariantInitialize.call``, kgcnn/layers/conv/painn_conv.py:261-290 (zeros/eps/ones/const/node)."""
    v = z.values
    if method == "zeros":
        out = np.zeros_like(v)
    elif method == "eps":
        out = np.zeros_like(v) + v.dtype.type(epsilon)
    elif method == "ones":
        out = np.ones_like(v)
    elif method == "const":
        out = np.ones_like(v) * v.dtype.type(value)
    elif method == "node":
        out = v
    else:
        raise ValueError("Unknown initialization method %s" % method)
    out = np.repeat(np.expand_dims(out, axis=1), dim, axis=1)
    return R(out, z.row_splits)


def split_embedding(r, num):
    """``SplitEmbedding``, kgcnn/layers/conv/painn_conv.py:329-340 (equal split of the last axis)."""
    return [R(x, r.row_splits) for x in np.split(r.values, num, axis=-1)]


def painn_conv(node, equivariant, rbf, envelope, r_ij, idx, p, act="swish", cutoff=None, conv_pool="sum"):
    """``PAiNNconv.call``, kgcnn/layers/conv/painn_conv.py:97-115."""
    s = dense(node, p["dense1/kernel"], p.get("dense1/bias"), act)
    s = dense(s, p["phi/kernel"], p.get("phi/bias"), "linear")
    s = gather_nodes_outgoing(s, idx)
    w = dense(rbf, p["w/kernel"], p.get("w/bias"), "linear")
    if cutoff is not None:
        w = lazy_multiply([w, envelope])
    sw = lazy_multiply([s, w])
    sw1, sw2, sw3 = split_embedding(sw, 3)
    ds = pooling_local_edges(node, sw1, idx, pooling_method=conv_pool)
    vj = gather_nodes_outgoing(equivariant, idx)
    sw2 = expand_dims(sw2, axis=-2)
    dv1 = lazy_multiply([sw2, vj])
    sw3 = expand_dims(sw3, axis=-2)
    r_ij = expand_dims(r_ij, axis=-1)
    dv2 = lazy_multiply([sw3, r_ij])
    dv = lazy_add([dv1, dv2])
    dv = pooling_local_edges(node, dv, idx, pooling_method=conv_pool)
    return ds, dv


def painn_update(node, equivariant, p, act="swish"):
    """``PAiNNUpdate.call``, kgcnn/layers/conv/painn_conv.py:201-214."""
    v_v = dense(equivariant, p["lin_v/kernel"], None, "linear")
    v_u = dense(equivariant, p["lin_u/kernel"], None, "linear")
    v_prod = scalar_product(v_u, v_v, axis=2)
    v_norm = euclidean_norm(v_v, axis=2)
    a = lazy_concatenate([node, v_norm], axis=-1)
    a = dense(a, p["dense1/kernel"], p.get("dense1/bias"), act)
    a = dense(a, p["a/kernel"], p.get("a/bias"), "linear")
    a_vv, a_sv, a_ss = split_embedding(a, 3)
    a_vv = expand_dims(a_vv, axis=-2)
    dv = lazy_multiply([a_vv, v_u])
    ds = lazy_multiply([v_prod, a_sv])
    ds = lazy_add([ds, a_ss])
    return ds, dv


def painn_forward(params, node_number, xyz, idx, depth=3, bessel_args=None, cutoff=None, equiv_method="zeros",
                  act="swish", conv_pool="sum", node_pooling="sum", output_mlp_act=("swish", "linear"),
                  return_intermediate=False):
    """``kgcnn.literature.PAiNN.make_model`` forward, kgcnn/literature/PAiNN.py:100-155 (graph output)."""
    bessel_args = bessel_args or {"num_radial": 20, "cutoff": 5.0, "envelope_exponent": 5}
    inter = {}
    z = embedding(node_number, params["embedding"])
    v = equivariant_initialize(z, dim=3, method=equiv_method)
    pos1, pos2 = node_position(xyz, idx)
    rij = edge_direction_normalized(pos1, pos2)
    d = node_distance_euclidean(pos1, pos2)
    env = cos_cutoff_envelope(d, cutoff)
    rbf = bessel_basis(d, frequencies=params.get("bessel/frequencies"), **bessel_args)
    inter["rbf"] = rbf.values
    inter["rij"] = rij.values
    for i in range(depth):
        ds, dv = painn_conv(z, v, rbf, env, rij, idx, _sub(params, "conv%d/" % i), act=act, cutoff=cutoff,
                            conv_pool=conv_pool)
        z = lazy_add([z, ds])
        v = lazy_add([v, dv])
        ds, dv = painn_update(z, v, _sub(params, "update%d/" % i), act=act)
        z = lazy_add([z, ds])
        v = lazy_add([v, dv])
        inter["z%d" % i] = z.values
        inter["v%d" % i] = v.values
    out = pooling_nodes(z, node_pooling)
    out = mlp(out, [(params["output_mlp/%d/kernel" % k], params.get("output_mlp/%d/bias" % k), output_mlp_act[k])
                    for k in range(len(output_mlp_act))])
    if return_intermediate:
        return out, inter
    return out


# ----------------------------------------------------------------------------------------
# kgcnn/layers/conv/gcn_conv.py, kgcnn/literature/GCN.py
# ----------------------------------------------------------------------------------------

def gcn_layer(node, edges, idx, p, act="relu", pooling_method="sum", normalize_by_weights=False,
              is_sorted=False, has_unconnected=True):
    """``GCN.call``, kgcnn/layers/conv/gcn_conv.py:85-90."""
    no = dense(node, p["kernel"], p.get("bias"), "linear")
    no = gather_nodes_outgoing(no, idx)
    nu = pooling_weighted_local_edges(node, no, idx, edges, pooling_method=pooling_method,
                                      normalize_by_weights=normalize_by_weights, is_sorted=is_sorted,
                                      has_unconnected=has_unconnected)
    return R(activation(act, nu.values), nu.row_splits)


def gcn_forward(params, node_attr, edge_weights, idx, depth=3, act="relu", pooling_method="sum",
                output_mlp_act=("relu", "relu", "softmax"), return_intermediate=False):
    """``kgcnn.literature.GCN.make_model`` forward with ``output_embedding='node'``,
    kgcnn/literature/GCN.py:95-109.  Returns the ragged node output's values."""
    inter = {}
    n = dense(node_attr, params["dense0/kernel"], params["dense0/bias"], "linear")
    for i in range(depth):
        n = gcn_layer(n, edge_weights, idx, _sub(params, "gcn%d/" % i), act=act, pooling_method=pooling_method)
        inter["n%d" % i] = n.values
    out = mlp(n, [(params["output_mlp/%d/kernel" % k], params.get("output_mlp/%d/bias" % k), output_mlp_act[k])
                  for k in range(len(output_mlp_act))])
    if return_intermediate:
        return out, inter
    return out


# ----------------------------------------------------------------------------------------
# kgcnn/layers/conv/gin_conv.py, gat_conv.py, dmpnn_conv.py (callers of the same primitives)
# ----------------------------------------------------------------------------------------

def gin_layer(node, idx, eps=0.0, pooling_method="sum"):
    """``GIN.call``, kgcnn/layers/conv/gin_conv.py:65-69."""
    ed = gather_nodes_outgoing(node, idx)
    nu = pooling_local_edges(node, ed, idx, pooling_method=pooling_method)
    no = (np.asarray(1, node.values.dtype) + np.asarray(eps, node.values.dtype)) * node.values
    return R(no + nu.values, node.row_splits)


def gine_layer(node, idx, edges, eps=0.0, pooling_method="sum", act="relu"):
    """``GINE.call``, kgcnn/layers/conv/gin_conv.py:147-153."""
    ed = gather_nodes_outgoing(node, idx)
    ed = R(activation(act, ed.values + edges.values), ed.row_splits)
    nu = pooling_local_edges(node, ed, idx, pooling_method=pooling_method)
    no = (np.asarray(1, node.values.dtype) + np.asarray(eps, node.values.dtype)) * node.values
    return R(no + nu.values, node.row_splits)


def attention_head_gat(node, edge, idx, p, act="kgcnn>leaky_relu", use_edge_features=False, use_final_activation=True):
    """``AttentionHeadGAT.call``, kgcnn/layers/conv/gat_conv.py:103-118.  ``p``: linear_trafo/{kernel,bias},
    alpha/kernel."""
    w_n = dense(node, p["linear_trafo/kernel"], p.get("linear_trafo/bias"), "linear")
    wn_in = gather_nodes_ingoing(w_n, idx)
    wn_out = gather_nodes_outgoing(w_n, idx)
    parts = [wn_in, wn_out, edge] if use_edge_features else [wn_in, wn_out]
    a_ij = dense(lazy_concatenate(parts, axis=-1), p["alpha/kernel"], None, act)
    h_i = pooling_local_edges_attention(node, wn_out, a_ij, idx)
    if use_final_activation:
        h_i = R(activation(act, h_i.values), h_i.row_splits)
    return h_i


def attention_head_gatv2(node, edge, idx, p, act="kgcnn>leaky_relu", use_edge_features=False,
                         use_final_activation=True):
    """``AttentionHeadGATV2.call``, kgcnn/layers/conv/gat_conv.py:200-218.  ``p``: linear_trafo/{kernel,bias},
    alpha_activation/{kernel,bias}, alpha/kernel."""
    w_n = dense(node, p["linear_trafo/kernel"], p.get("linear_trafo/bias"), "linear")
    n_in = gather_nodes_ingoing(node, idx)
    n_out = gather_nodes_outgoing(node, idx)
    wn_out = gather_nodes_outgoing(w_n, idx)
    parts = [n_in, n_out, edge] if use_edge_features else [n_in, n_out]
    a_ij = dense(lazy_concatenate(parts, axis=-1), p["alpha_activation/kernel"], p.get("alpha_activation/bias"), act)
    a_ij = dense(a_ij, p["alpha/kernel"], None, "linear")
    h_i = pooling_local_edges_attention(node, wn_out, a_ij, idx)
    if use_final_activation:
        h_i = R(activation(act, h_i.values), h_i.row_splits)
    return h_i


def gat_forward(params, node_attr, edge_attr, idx, depth=3, heads=5, concat_heads=False, v2=False,
                act="kgcnn>leaky_relu", use_edge_features=True, use_final_activation=False,
                pooling_method="mean", output_mlp_act=("relu", "relu", "sigmoid")):
    """``kgcnn.literature.GAT.make_model`` / ``GATv2.make_model`` forward with feature inputs and
    ``output_embedding='graph'`` (kgcnn/literature/GAT.py:89-112).  ``params``: dense0/{kernel,bias},
    block{i}/head{h}/<head params>, output_mlp/{k}/{kernel,bias}."""
    head_fn = attention_head_gatv2 if v2 else attention_head_gat
    nk = dense(node_attr, params["dense0/kernel"], params.get("dense0/bias"), "linear")
    for i in range(depth):
        outs = [head_fn(nk, edge_attr, idx, _sub(params, "block%d/head%d/" % (i, h)), act=act,
                        use_edge_features=use_edge_features, use_final_activation=use_final_activation)
                for h in range(heads)]
        if concat_heads:
            nk = lazy_concatenate(outs, axis=-1)
        else:
            acc = outs[0].values
            for o in outs[1:]:
                acc = acc + o.values
            mean = acc / np.asarray(len(outs), acc.dtype)      # LazyAverage, kgcnn/layers/modules.py
            nk = R(activation(act, mean), outs[0].row_splits)
    pooled = pooling_nodes(nk, pooling_method=pooling_method)
    return mlp(pooled, [(params["output_mlp/%d/kernel" % k], params.get("output_mlp/%d/bias" % k), output_mlp_act[k])
                        for k in range(len(output_mlp_act))])


def dmpnn_gather_edges_pairs(edges, pair_index):
    """``DMPNNGatherEdgesPairs.call``, kgcnn/layers/conv/dmpnn_conv.py:39-46: reverse-edge rows, zeros where the pair
    index is negative."""
    pairs = np.asarray(pair_index.values)
    safe = R(np.where(pairs >= 0, pairs, np.zeros_like(pairs)), pair_index.row_splits)
    gathered = gather_nodes_ingoing(edges, safe)
    keep = (pairs[:, 0] >= 0).reshape((-1,) + (1,) * (gathered.values.ndim - 1))
    return R(np.where(keep, gathered.values, np.zeros_like(gathered.values)), gathered.row_splits)


def dmpnn_pooling_edges_directed(nodes, edges, idx, reverse_pair):
    """``DMPNNPPoolingEdgesDirected.call``, kgcnn/layers/conv/dmpnn_conv.py:83-87."""
    pool_edge_receive = pooling_local_edges(nodes, edges, idx, pooling_method="sum")
    ed_new = gather_nodes_outgoing(pool_edge_receive, idx)
    ed_not = dmpnn_gather_edges_pairs(edges, reverse_pair)
    return R(ed_new.values - ed_not.values, ed_new.row_splits)


def layer_normalization(x, gamma=None, beta=None, epsilon=1e-3):
    """Keras ``LayerNormalization`` over the last axis, as ``GraphLayerNormalization`` applies it to the values
    (kgcnn/layers/norm.py:60-63, 94-105): moments with the variance as mean squared difference, then the
    ``tf.nn.batch_normalization`` form ``x * inv + (beta - mean * inv)``, ``inv = rsqrt(var + eps) * gamma``."""
    x = np.asarray(x)
    mean = np.mean(x, axis=-1, keepdims=True, dtype=x.dtype)
    var = np.mean(np.square(x - mean), axis=-1, keepdims=True, dtype=x.dtype)
    inv = (np.asarray(1, x.dtype) / np.sqrt(var + np.asarray(epsilon, x.dtype))).astype(x.dtype)
    if gamma is not None:
        inv = inv * gamma
    shift = -mean * inv
    if beta is not None:
        shift = beta + shift
    return (x * inv + shift).astype(x.dtype)


def graph_sage_node_layer(node, idx, p, edge=None, act="relu", pooling_method="sum"):
    """``GraphSageNodeLayer.call``, kgcnn/layers/conv/sage_conv.py:83-100.  ``p``: nb/{kernel,bias}, self/{kernel,bias},
    norm/{gamma,beta}."""
    msg = gather_nodes_outgoing(node, idx)
    if edge is not None:
        msg = lazy_concatenate([msg, edge], axis=-1)
    msg = dense(msg, p["nb/kernel"], p.get("nb/bias"), act)
    nu = pooling_local_edges(node, msg, idx, pooling_method=pooling_method)
    n = dense(lazy_concatenate([node, nu], axis=-1), p["self/kernel"], p.get("self/bias"), act)
    return R(layer_normalization(n.values, p.get("norm/gamma"), p.get("norm/beta")), n.row_splits)


def graph_sage_edge_update_layer(node, edge, idx, p, act="relu", use_normalization=True):
    """``GraphSageEdgeUpdateLayer.call``, kgcnn/layers/conv/sage_conv.py:179-185."""
    pair = gather_nodes(node, idx)
    ed = dense(lazy_concatenate([edge, pair], axis=-1), p["mlp/kernel"], p.get("mlp/bias"), act)
    if use_normalization:
        ed = R(layer_normalization(ed.values, p.get("norm/gamma"), p.get("norm/beta")), ed.row_splits)
    return ed


def dmpnn_forward(params, node_attr, edge_attr, idx, reverse_pair, depth=5, pooling_method="sum",
                  output_mlp_act=("relu", "relu", "linear")):
    """``kgcnn.literature.DMPNN.make_model`` forward with feature inputs, ``output_embedding='graph'`` and inference-mode
    dropout (kgcnn/literature/DMPNN.py:132-152).  ``params``: h0/, edge/, node/ {kernel,bias}, output_mlp/{k}/..."""
    h_n0 = gather_nodes_outgoing(node_attr, idx)
    h0 = dense(lazy_concatenate([h_n0, edge_attr], axis=-1), params["h0/kernel"], params.get("h0/bias"), "relu")
    h = h0
    for _ in range(depth):
        m_vw = dmpnn_pooling_edges_directed(node_attr, h, idx, reverse_pair)
        h = dense(m_vw, params["edge/kernel"], params.get("edge/bias"), "linear")
        h = R(activation("relu", h.values + h0.values), h.row_splits)
    mv = pooling_local_edges(node_attr, h, idx, pooling_method=pooling_method)
    hv = dense(lazy_concatenate([mv, node_attr], axis=-1), params["node/kernel"], params.get("node/bias"), "relu")
    out = pooling_nodes(hv, pooling_method=pooling_method)
    return mlp(out, [(params["output_mlp/%d/kernel" % k], params.get("output_mlp/%d/bias" % k), output_mlp_act[k])
                     for k in range(len(output_mlp_act))])


def gin_forward(params, node_attr, idx, depth=3, gin_mlp_act=("relu", "linear"), last_mlp_act=("relu", "relu", "linear"),
                output_act="softmax"):
    """``kgcnn.literature.GIN.make_model`` forward with feature inputs, ``output_embedding='graph'``, no normalisation,
    inference-mode dropout (kgcnn/literature/GIN.py:89-102).  ``params``: dense0/, gin{i}/eps, mlp{i}/{k}/,
    last{j}/{k}/ (j = 0..depth), output/."""
    n = dense(node_attr, params["dense0/kernel"], params.get("dense0/bias"), "linear")
    embeddings = [n]
    for i in range(depth):
        n = gin_layer(n, idx, eps=params.get("gin%d/eps" % i, 0.0))
        n = mlp(n, [(params["mlp%d/%d/kernel" % (i, k)], params.get("mlp%d/%d/bias" % (i, k)), a)
                    for k, a in enumerate(gin_mlp_act)])
        embeddings.append(n)
    total = None
    for j, x in enumerate(embeddings):
        pooled = pooling_nodes(x, pooling_method="mean")
        part = mlp(pooled, [(params["last%d/%d/kernel" % (j, k)], params.get("last%d/%d/bias" % (j, k)), a)
                            for k, a in enumerate(last_mlp_act)])
        total = part if total is None else total + part
    return mlp(total, [(params["output/kernel"], params.get("output/bias"), output_act)])


def graphsage_forward(params, node_attr, edge_attr, idx, depth=3, use_edge_features=True, pooling_method="segment_mean",
                      mlp_act=("relu", "linear"), pooling_nodes_method="mean", output_mlp_act=("relu", "relu", "sigmoid")):
    """``kgcnn.literature.GraphSAGE.make_model`` forward with feature inputs and ``output_embedding='graph'``
    (kgcnn/literature/GraphSAGE.py:103-123).  ``params``: block{i}/edge/{k}/{kernel,bias}, block{i}/node/{k}/...,
    block{i}/norm/{gamma,beta}, output_mlp/{k}/..."""
    n = node_attr
    for i in range(depth):
        eu = gather_nodes_outgoing(n, idx)
        if use_edge_features:
            eu = lazy_concatenate([eu, edge_attr], axis=-1)
        eu = mlp(eu, [(params["block%d/edge/%d/kernel" % (i, k)], params.get("block%d/edge/%d/bias" % (i, k)), a)
                      for k, a in enumerate(mlp_act)])
        nu = pooling_local_edges(n, eu, idx, pooling_method=pooling_method)
        nu = lazy_concatenate([n, nu], axis=-1)
        n = mlp(nu, [(params["block%d/node/%d/kernel" % (i, k)], params.get("block%d/node/%d/bias" % (i, k)), a)
                     for k, a in enumerate(mlp_act)])
        n = R(layer_normalization(n.values, params["block%d/norm/gamma" % i], params["block%d/norm/beta" % i]),
              n.row_splits)
    out = pooling_nodes(n, pooling_method=pooling_nodes_method)
    return mlp(out, [(params["output_mlp/%d/kernel" % k], params.get("output_mlp/%d/bias" % k), output_mlp_act[k])
                     for k in range(len(output_mlp_act))])


def megnet_block(node, edge, idx, env, p, act="kgcnn>softplus2", pooling_method="mean"):
    """``MEGnetBlock.call``, kgcnn/layers/conv/megnet_conv.py:96-120.  ``env`` is a dense ``(G, Fu)`` array; ``p`` holds
    phi_e{,_1,_2}/phi_n{,_1,_2}/phi_u{,_1,_2} kernels and biases.  Returns ``(nodes, edges, env)``."""
    def chain(x, name):
        for k, a in (("", act), ("_1", act), ("_2", "linear")):
            x = dense_values(x, p["%s%s/kernel" % (name, k)], p.get("%s%s/bias" % (name, k)), a)
        return x
    e_n = gather_nodes(node, idx)
    e_u = gather_state(env, edge)
    ep = R(chain(np.concatenate([e_n.values, edge.values, e_u.values], axis=-1), "phi_e"), edge.row_splits)
    vb = pooling_local_edges(node, ep, idx, pooling_method=pooling_method)
    v_u = gather_state(env, node)
    vp = R(chain(np.concatenate([vb.values, node.values, v_u.values], axis=-1), "phi_n"), node.row_splits)
    es = pooling_nodes(ep, pooling_method=pooling_method)
    vs = pooling_nodes(vp, pooling_method=pooling_method)
    up = chain(np.concatenate([es, vs, env], axis=-1), "phi_u")
    return vp, ep, up


# ----------------------------------------------------------------------------------------
# kgcnn/layers/pool/set2set.py, kgcnn/layers/conv/mpnn_conv.py, kgcnn/literature/{Megnet,NMPN}.py
# parity unpinned: the reference holds no value fixture for these; the LSTM / GRU arithmetic is Keras' documented cell
# math (keras/layers/rnn/{lstm,gru}.py: gate order i,f,c,o resp. z,r,h; GRU reset_after=True), restated here.
# ----------------------------------------------------------------------------------------

def pooling_set2set(r, kernel, bias, T=3, pooling_method="mean", init_qstar="mean", act="tanh", rec_act="sigmoid"):
    """``PoolingSet2Set.call``, kgcnn/layers/pool/set2set.py:172-199 (+ ``init_qstar_mean`` :245-263).  The reference
    calls a stateless Keras LSTM on a length-1 sequence in every round: one LSTM step from h0 = c0 = 0, so
    ``q = o * act(i * act(z_c))`` with ``z = q* kernel + bias`` (the recurrent kernel multiplies h0 = 0).
    Returns ``(G, 1, 2 * channels)``."""
    m = np.asarray(r.values)
    dt = m.dtype
    ids = value_rowids(r)
    lens = row_lengths(r)
    g, f = len(lens), m.shape[1]
    pool = (lambda x: x.mean(axis=1)) if pooling_method == "mean" else (lambda x: x.sum(axis=1))

    def attend(q):
        et = pool(m * np.repeat(q, lens, axis=0))                              # f_et, :201-213
        mx = np.full(g, -np.inf, dt)
        np.maximum.at(mx, ids, et)                                             # segment_max, :221-225
        at = np.exp(et - np.repeat(mx, lens))
        norm = np.zeros(g, dt)
        np.add.at(norm, ids, at)                                               # segment_sum
        with np.errstate(divide="ignore"):
            inv = np.where(norm == 0, dt.type(0), dt.type(1) / norm)           # reciprocal_no_nan, :231
        at = np.repeat(inv, lens) * at
        rt = np.zeros((g, f), dt)
        for n in range(m.shape[0]):                                            # segment_sum in row order
            rt[ids[n]] += m[n] * at[n]
        return rt

    if init_qstar == "mean":
        q = np.zeros((g, f), dt)
        for n in range(m.shape[0]):
            q[ids[n]] += m[n]
        q = q / np.maximum(lens, 1).astype(dt)[:, None]
        qstar = np.concatenate([q, attend(q)], axis=1)
    else:
        qstar = np.zeros((g, 2 * f), dt)
    for _ in range(T):
        z = np.matmul(qstar, kernel.astype(dt)) + (bias.astype(dt) if bias is not None else 0)
        zi, _, zc, zo = np.split(z, 4, axis=1)
        c = activation(rec_act, zi) * activation(act, zc)
        q = (activation(rec_act, zo) * activation(act, c)).astype(dt)
        qstar = np.concatenate([q, attend(q)], axis=1)
    return qstar[:, None, :]


def gru_update(nodes, updates, kernel, recurrent_kernel, bias, act="tanh", rec_act="sigmoid"):
    """``GRUUpdate.call``, kgcnn/layers/conv/mpnn_conv.py:183-200: one Keras GRUCell step (reset_after=True) with the node
    values as state and the pooled messages as input; ``bias`` is Keras' ``(2, 3u)`` (input row, recurrent row)."""
    h, x = np.asarray(nodes.values), np.asarray(updates.values)
    dt = h.dtype
    mx = np.matmul(x, kernel.astype(dt)) + (bias[0].astype(dt) if bias is not None else 0)
    mh = np.matmul(h, recurrent_kernel.astype(dt)) + (bias[1].astype(dt) if bias is not None else 0)
    xz, xr, xh = np.split(mx, 3, axis=1)
    hz, hr, hh = np.split(mh, 3, axis=1)
    z = activation(rec_act, xz + hz)
    rg = activation(rec_act, xr + hr)
    cand = activation(act, xh + rg * hh)
    return R((z * h + (dt.type(1) - z) * cand).astype(dt), nodes.row_splits)


def trafo_edge_net_messages(edges, kernel, bias, target_shape, act="linear"):
    """``TrafoEdgeNetMessages.call``, mpnn_conv.py:44-57: Dense + reshape to ``(M, F', F)``."""
    up = dense_values(edges.values, kernel, bias, act)
    return R(up.reshape(up.shape[0], int(target_shape[0]), int(target_shape[1])), edges.row_splits)


def matmul_messages(trafo, edges):
    """``MatMulMessages.call``, mpnn_conv.py:88-103: ``batch_dot`` of per-edge matrices with per-edge vectors."""
    return R(np.einsum("mrc,mc->mr", trafo.values, edges.values).astype(edges.values.dtype), edges.row_splits)


def megnet_forward(weights, node_number, xyz, idx, env_number, nblocks=3, has_ff=True, use_set2set=True, gauss_args=None,
                   set2set_args=None, act="kgcnn>softplus2", ff_act="kgcnn>softplus2",
                   output_act=("kgcnn>softplus2", "kgcnn>softplus2", "linear"), ff_layers=2):
    """``kgcnn.literature.Megnet.make_model`` forward, kgcnn/literature/Megnet.py:117-190, with the weights as the flat list
    ``model.get_weights()`` gives (construction order: embeddings, feed-forward MLPs, blocks, readouts, output MLP)."""
    gauss_args = gauss_args or {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4}
    set2set_args = set2set_args or {"channels": 16, "T": 3, "pooling_method": "sum", "init_qstar": "0"}
    w = iter(weights)
    take = lambda k: [next(w) for _ in range(k)]

    def take_mlp(n_layers, acts):
        return [(next(w), next(w), a) for _, a in zip(range(n_layers), acts)]

    emb_n, emb_u = take(2)
    n = embedding(node_number, emb_n)
    u = emb_u[np.asarray(env_number).astype(np.int32)]
    pos1, pos2 = node_position(xyz, idx)
    ed = gauss_basis(node_distance_euclidean(pos1, pos2), **gauss_args)

    def ff_trio():
        return [take_mlp(ff_layers, [ff_act] * ff_layers) for _ in range(3)]

    def block_params():
        p = {}
        for name in ("phi_n", "phi_e", "phi_u"):                  # attribute order of MEGnetBlock
            for suffix in ("", "_1", "_2"):
                p["%s%s/kernel" % (name, suffix)], p["%s%s/bias" % (name, suffix)] = next(w), next(w)
        return p

    trio = ff_trio()
    vp, ep, up = mlp(n, trio[0]), mlp(ed, trio[1]), mlp(u, trio[2])
    vp2, ep2, up2 = vp, ep, up
    for i in range(nblocks):
        if has_ff and i > 0:
            trio = ff_trio()
            vp2, ep2, up2 = mlp(vp, trio[0]), mlp(ep, trio[1]), mlp(up, trio[2])
        vp2, ep2, up2 = megnet_block(vp2, ep2, idx, up2, block_params(), act=act)
        vp, ep, up = lazy_add([vp2, vp]), lazy_add([ep2, ep]), up2 + up
    if use_set2set:
        kv, bv, ke, be = take(4)
        s2s = {k: set2set_args[k] for k in ("T", "pooling_method", "init_qstar")}
        lk_v, _, lb_v = take(3)
        lk_e, _, lb_e = take(3)
        vs = pooling_set2set(dense(vp, kv, bv, "linear"), lk_v, lb_v, **s2s)
        es = pooling_set2set(dense(ep, ke, be, "linear"), lk_e, lb_e, **s2s)
    else:
        vs, es = pooling_nodes(vp, "mean"), pooling_nodes(ep, "mean")
    final = np.concatenate([vs.reshape(vs.shape[0], -1), es.reshape(es.shape[0], -1), up], axis=-1)
    return mlp(final, take_mlp(len(output_act), output_act))


def nmpn_forward(weights, node_number, edge_number, idx, depth=3, node_dim=64, set2set_args=None, edge_act="swish",
                 edge_layers=3, output_act=("selu", "selu", "sigmoid"), output_bias=(True, True, False),
                 pooling_method="sum"):
    """``kgcnn.literature.NMPN.make_model`` forward (graph output, embedded node / edge numbers, Set2Set readout),
    kgcnn/literature/NMPN.py:110-170, weights as the flat ``model.get_weights()`` list."""
    set2set_args = set2set_args or {"channels": 32, "T": 3, "pooling_method": "sum", "init_qstar": "0"}
    w = iter(weights)
    emb_n, emb_e = next(w), next(w)
    n0 = embedding(node_number, emb_n)
    ed = embedding(edge_number, emb_e)
    n = dense(n0, next(w), next(w), "linear")
    nets = []
    for _ in range(2):                                            # edge network "in", then "out"
        layers = [(next(w), next(w), edge_act) for _ in range(edge_layers)]
        nets.append(trafo_edge_net_messages(mlp(ed, layers), next(w), next(w), (node_dim, node_dim)))
    gru_k, gru_r, gru_b = next(w), next(w), next(w)
    for _ in range(depth):
        m_in = matmul_messages(nets[0], gather_nodes_outgoing(n, idx))
        m_out = matmul_messages(nets[1], gather_nodes_ingoing(n, idx))
        eu = pooling_local_edges(n, lazy_concatenate([m_in, m_out], axis=-1), idx, pooling_method=pooling_method)
        n = gru_update(n, eu, gru_k, gru_r, gru_b)
    n = lazy_concatenate([n0, n], axis=-1)
    out = dense(n, next(w), next(w), "linear")
    lk, _, lb = next(w), next(w), next(w)
    out = pooling_set2set(out, lk, lb, **{k: set2set_args[k] for k in ("T", "pooling_method", "init_qstar")})
    out = out.reshape(out.shape[0], -1)
    return mlp(out, [(next(w), next(w) if b else None, a) for a, b in zip(output_act, output_bias)])


# ----------------------------------------------------------------------------------------
# kgcnn/layers/casting.py
# ----------------------------------------------------------------------------------------

def ragged_to_padded(r, default_value=0):
    """``ChangeTensorType(ragged -> padded/mask)``, kgcnn/layers/casting.py:79-84: ``(padded, mask)``."""
    lens = row_lengths(r)
    g = len(lens)
    nmax = int(lens.max()) if g > 0 else 0
    padded = np.full((g, nmax) + r.values.shape[1:], default_value, dtype=r.values.dtype)
    mask = np.zeros((g, nmax) + r.values.shape[1:], dtype=r.values.dtype)
    for i in range(g):
        padded[i, :lens[i]] = r.values[r.row_splits[i]:r.row_splits[i + 1]]
        mask[i, :lens[i]] = 1
    return padded, mask


def to_dtype(tree, dtype):
    """Cast every floating array of a dict / R / list to ``dtype`` (float64 twin for error budgets)."""
    if isinstance(tree, dict):
        return {k: to_dtype(v, dtype) for k, v in tree.items()}
    if isinstance(tree, R):
        return R(to_dtype(tree.values, dtype), tree.row_splits)
    if isinstance(tree, (list, tuple)):
        return type(tree)(to_dtype(v, dtype) for v in tree)
    a = np.asarray(tree)
    if np.issubdtype(a.dtype, np.floating):
        return a.astype(dtype)
    return a
