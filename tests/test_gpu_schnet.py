"""SchNet forward (kgcnn.literature.Schnet.make_model) on the HIP engine vs the CPU oracle.

Tolerance: 1e-5 of the output scale (BASELINE.json north_star), with the float64 twin of the oracle as the error
budget: the engine must be as close to float64 truth as the float32 oracle is (within a factor), which is the
meaningful statement when two float32 pipelines round differently."""
import os

import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth
from oracle import kgcnn_oracle as ko

pytestmark = pytest.mark.gpu


def _inputs(b):
    from gcnn_keras_amd.ragged import RaggedTensor
    return [RaggedTensor.from_numpy(b["node_number"], b["node_splits"]),
            RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
            RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]


def _oracle(p, b, depth, dtype=np.float32):
    pp = ko.to_dtype(p, dtype)
    return ko.schnet_forward(pp, ko.R(b["node_number"], b["node_splits"]),
                             ko.R(b["node_coordinates"].astype(dtype), b["node_splits"]),
                             ko.R(b["edge_indices"], b["edge_splits"]), depth=depth)


def _check(got, ref32, ref64):
    scale = float(np.max(np.abs(ref64)))
    err_engine = float(np.max(np.abs(got - ref64)))
    err_oracle = float(np.max(np.abs(ref32 - ref64)))
    assert np.max(np.abs(got - ref32)) <= 1e-5 * scale, (np.max(np.abs(got - ref32)), scale)
    assert err_engine <= max(4 * err_oracle, 2e-6 * scale), (err_engine, err_oracle, scale)


@pytest.mark.parametrize("num_graphs,seed", [(6, 11), (1, 5), (128, 1234)])
def test_schnet_layerwise_forward(num_graphs, seed, golden_dir):
    from gcnn_keras_amd.literature import Schnet
    b = synth.qm9_like_batch(num_graphs=num_graphs, seed=seed)
    p = synth.schnet_params(seed=7, random_bias=True)
    model = Schnet.make_model(depth=3)
    model.set_weights(list(p.values()))
    out = model(_inputs(b)).cpu().numpy()
    assert out.shape == (num_graphs, 1)
    _check(out, _oracle(p, b, 3), _oracle(p, b, 3, np.float64))
    if (num_graphs, seed) == (6, 11):
        frozen = np.load(os.path.join(golden_dir, "frozen_schnet_small.npz"))["out"]
        assert np.max(np.abs(out - frozen)) <= 1e-5 * np.max(np.abs(frozen))


def test_schnet_config_surface():
    from gcnn_keras_amd.literature import Schnet
    with pytest.raises(ValueError):
        Schnet.make_model(not_a_kwarg=1)                 # update_model_kwargs rejects unknown keys
    m = Schnet.make_model()                              # model_default depth is 4 (kgcnn/literature/Schnet.py:34)
    names = [n for n, _ in m.weights]
    assert len(names) == 1 + 2 + 4 * 9 + 4 + 4
    shapes = [tuple(t.shape) for _, t in m.weights]
    assert shapes[0] == (95, 64) and shapes[1] == (64, 128)
    assert shapes[3] == (20, 128) and shapes[7] == (128, 128)   # cfconv.dense1 kernel, interaction.dense1 (no bias)
    cfg = m.layers[2].get_config()
    assert cfg["units"] == 128 and cfg["cfconv_pool"] == "sum" and cfg["activation"] == "kgcnn>shifted_softplus"
