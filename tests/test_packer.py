"""Host batch packer vs NumPy / the oracle's partition arithmetic.  CPU only: the native packer runs
on host pointers; the copies to the device are covered by tests/test_gpu_packer.py."""
import numpy as np
import pytest

from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.data.packer import BatchPacker, HostBuffer, pack_edge_index, pack_rows
from oracle import kgcnn_oracle as ko


def _split(b, key, splits_key):
    s = b[splits_key]
    return [b[key][s[i]:s[i + 1]] for i in range(len(s) - 1)]


@pytest.mark.parametrize("src,dst", [("float64", "float32"), ("float32", None), ("int32", "int64"), ("int64", None),
                                     ("float32", "float64"), ("int64", "float32"), ("int16", "int64")])
def test_pack_rows_equals_concatenate(src, dst):
    # the reference: np.concatenate(numpy_list, axis=0, dtype=dtype) + row lengths, kgcnn/data/utils.py:156-157
    rng = np.random.default_rng(3)
    lens = [4, 0, 7, 1, 0, 3]
    arrays = [(rng.normal(size=(n, 2, 3)) * 50).astype(src) for n in lens]
    values, splits = pack_rows(arrays, dtype=dst)
    want = np.concatenate(arrays, axis=0, dtype=dst)
    assert values.dtype == want.dtype and values.shape == want.shape
    assert np.array_equal(values, want)
    assert np.array_equal(splits, np.concatenate([[0], np.cumsum(lens)]))


def test_pack_rows_edge_cases():
    v, s = pack_rows([], dtype="float32")
    assert v.shape[0] == 0 and np.array_equal(s, [0])
    v, s = pack_rows([np.zeros((0, 3)), np.zeros((0, 3))], dtype="float32")
    assert v.shape == (0, 3) and np.array_equal(s, [0, 0, 0])
    with pytest.raises(ValueError):
        pack_rows([np.zeros((2, 3)), np.zeros((2, 4))])
    with pytest.raises(TypeError):
        pack_rows([np.zeros((2, 3))], dtype="float16")
    with pytest.raises(_ffi.EngineError):     # float -> int is not a conversion the packer offers (MP_ENOTSUP)
        pack_rows([np.zeros((2, 3), dtype=np.float32)], dtype="int64")


def test_pack_rows_threads_agree():
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 400, size=300)
    arrays = [rng.normal(size=(n, 128)).astype(np.float32) for n in lens]   # 30 MB: takes the threaded branch
    a, sa = pack_rows(arrays, threads=1)
    b, sb = pack_rows(arrays, threads=6)
    assert np.array_equal(a, b) and np.array_equal(sa, sb)
    assert np.array_equal(a, np.concatenate(arrays))


def test_host_buffer_reuse_and_growth():
    hb = HostBuffer(pinned=False)
    a = hb.view((16,), np.int32)
    a[:] = 5
    b = hb.view((8,), np.int32)            # same block, no reallocation
    assert np.array_equal(b, np.full(8, 5))
    c = hb.view((1 << 16,), np.float64)    # grows
    c[:] = 1.0
    assert c.sum() == float(1 << 16)


def _oracle_plan(idx_list, node_counts):
    lens = [len(x) for x in idx_list]
    K = 2
    idx = np.concatenate([np.asarray(x, dtype=np.int64).reshape(-1, K) for x in idx_list]) if sum(lens) else \
        np.zeros((0, K), np.int64)
    es = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ns = np.concatenate([[0], np.cumsum(node_counts)]).astype(np.int64)
    shifted = ko.partition_row_indexing(idx, ns, es, "row_splits", "row_splits", from_indexing="sample",
                                        to_indexing="batch")
    return idx, es, ns, shifted


@pytest.mark.parametrize("num_graphs,seed", [(1, 2), (9, 4), (128, 1234)])
def test_pack_edge_index_matches_partition_shift(num_graphs, seed):
    b = synth.qm9_like_batch(num_graphs=num_graphs, seed=seed)
    idx_list = _split(b, "edge_indices", "edge_splits")
    node_counts = np.diff(b["node_splits"])
    for dtype in (np.int64, np.int32):
        plan = pack_edge_index([x.astype(dtype) for x in idx_list], node_counts, threads=3)
        idx, es, ns, shifted = _oracle_plan(idx_list, node_counts)
        M, N = plan["M"], plan["N"]
        assert (M, N) == (len(idx), int(ns[-1]))
        assert np.array_equal(plan["idx"], idx) and plan["idx"].dtype == np.int64      # the API tensor, untouched
        assert np.array_equal(plan["edge_splits"], es) and np.array_equal(plan["node_splits"], ns)
        assert np.array_equal(plan["cols"][:, :M].T, shifted)                          # kgcnn/ops/partition.py:140-155
        assert plan["flags"] == _ffi.MP_FLAG_UNSORTED_COL1       # SetRange edges: receiver-sorted, senders not
        want_ptr = np.searchsorted(shifted[:, 0], np.arange(N + 1), side="left")
        assert np.array_equal(plan["csr"], want_ptr)


def test_pack_edge_index_flags_and_empties():
    # unsorted receivers: flag set, CSR left at zeros (the device sorts in that case)
    plan = pack_edge_index([np.array([[1, 0], [0, 1]]), np.zeros((0, 2), np.int64), np.array([[0, 0]])], [2, 0, 1])
    assert plan["flags"] & _ffi.MP_FLAG_UNSORTED_COL0
    assert not plan["flags"] & _ffi.MP_FLAG_OOB
    assert np.array_equal(plan["csr"], np.zeros(4, np.int32))
    assert np.array_equal(plan["cols"][:, :3], [[1, 0, 2], [0, 1, 2]])
    # out-of-range index: flagged and clamped like mp_index_prepare_i64, the API tensor keeps the raw value
    plan = pack_edge_index([np.array([[0, 5], [1, -1]])], [2])
    assert plan["flags"] & _ffi.MP_FLAG_OOB
    assert np.array_equal(plan["cols"][:, :2], [[0, 1], [1, 0]])
    assert np.array_equal(plan["idx"], [[0, 5], [1, -1]])
    # no graphs / no edges
    plan = pack_edge_index([], [])
    assert plan["M"] == 0 and plan["N"] == 0 and plan["flags"] == 0
    plan = pack_edge_index([np.zeros((0, 2), np.int64)], [4])
    assert plan["M"] == 0 and np.array_equal(plan["csr"], np.zeros(5, np.int32))
    with pytest.raises(ValueError):
        pack_edge_index([np.zeros((1, 2), np.int64)], [1, 2])


def test_batch_packer_host_half_follows_memory_graph_list_items():
    # MemoryGraphList.tensor(items): ragged items -> concatenate + lengths, others -> np.array(props)
    # (kgcnn/data/base.py:203-217)
    b = synth.qm9_like_batch(num_graphs=7, seed=21)
    graphs = [{"node_number": z, "node_coordinates": xyz, "edge_indices": ei, "graph_labels": np.array([float(i)])}
              for i, (z, xyz, ei) in enumerate(zip(_split(b, "node_number", "node_splits"),
                                                   _split(b, "node_coordinates", "node_splits"),
                                                   _split(b, "edge_indices", "edge_splits")))]
    items = [{"name": "node_number", "ragged": True, "dtype": "float32"},
             {"name": "node_coordinates", "ragged": True, "dtype": "float32"},
             {"name": "edge_indices", "ragged": True, "dtype": "int64"},
             {"name": "graph_labels", "ragged": False, "dtype": "float32"}]
    packer = BatchPacker(items, index_item="edge_indices", node_item="node_number")
    host = packer.pack_host(graphs)
    assert np.array_equal(host["node_number"][0], b["node_number"])
    assert np.array_equal(host["node_coordinates"][0], b["node_coordinates"])
    assert np.array_equal(host["node_coordinates"][1], b["node_splits"])
    assert np.array_equal(host["edge_indices"][0], b["edge_indices"])
    assert np.array_equal(host["edge_indices"][1], b["edge_splits"])
    assert host["graph_labels"].shape == (7, 1) and host["graph_labels"].dtype == np.float32
    assert host["__plan__"]["N"] == len(b["node_number"])


def test_memory_graph_list_and_output_translation_host_side():
    # list behaviour of the reference container (kgcnn/data/base.py:28-150) and the predictor's output mapping
    # (kgcnn/moldyn/base.py:81-104); nothing here touches the device
    from gcnn_keras_amd.data.base import GraphDict, MemoryGraphList
    from gcnn_keras_amd.moldyn import MolDynamicsModelPredictor
    gl = MemoryGraphList([{"node_number": np.array([1, 6]), "energy": np.array([0.5])}, {"node_number": np.array([8])}])
    assert len(gl) == 2 and isinstance(gl[0], GraphDict)
    assert [None if v is None else v.tolist() for v in gl.obtain_property("energy")] == [[0.5], None]
    cp = gl.copy()
    cp[0].set("energy", [2.0])
    assert gl[0]["energy"][0] == 0.5 and cp[0]["energy"][0] == 2.0
    gl[1].apply_preprocessor(lambda g: {"charge": np.array([len(g["node_number"])])})
    assert gl[1]["charge"].tolist() == [1]
    with pytest.raises(TypeError):
        gl[1].apply_preprocessor("set_range")      # serialized names are not resolved on this engine
    tr = MolDynamicsModelPredictor._translate_properties
    assert tr([1, 2], ["energy", "forces"]) == {"energy": 1, "forces": 2}
    assert tr({"e": 1, "f": 2}, {"energy": "e", "forces": "f"}) == {"energy": 1, "forces": 2}
    assert tr(3.0, "energy") == {"energy": 3.0}
    with pytest.raises(TypeError):
        tr(3.0, 7)
    with pytest.raises(TypeError):
        MolDynamicsModelPredictor(model=None, graph_preprocessors=[{"class_name": "SetRange"}])
