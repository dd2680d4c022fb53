"""MD inference driver (mirror of kgcnn/moldyn/base.py:106-165): eager and HIP-graph-replayed energy + force calls.
What the predictor returns is held to the ORACLE - energies to the NumPy restatement (float32 + float64 twin), forces of every
atom to the analytic reference oracle/torch_force_oracle.py - at every MD step, not to another call of the same model."""
import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth
from helpers import painn_weight_list
from oracle import kgcnn_oracle as ko
from oracle import torch_force_oracle as tfo
from parity import assert_forces_close, assert_rows_close

P = synth.painn_params(seed=8, random_bias=True)

pytestmark = pytest.mark.gpu

ITEMS = [{"name": "node_number", "ragged": True, "dtype": "float32"},
         {"name": "node_coordinates", "ragged": True, "dtype": "float32"},
         {"name": "range_indices", "ragged": True, "dtype": "int64"}]


def _painn_ef():
    from gcnn_keras_amd.literature import PAiNN
    from gcnn_keras_amd.model.force import EnergyForceModel
    energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
    energy.set_weights(painn_weight_list(P))
    return EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_as_dict=True,
                            output_to_tensor=False, output_squeeze_states=True)


def _check_against_oracle(out, b, xyz=None, what=""):
    """Predictor output (list of per-graph dicts) against the oracle's energies and analytic forces for batch ``b``."""
    bb = dict(b) if xyz is None else dict(b, node_coordinates=np.asarray(xyz, np.float32))
    e_ref = [ko.painn_forward(ko.to_dtype(P, dt), ko.R(bb["node_number"], bb["node_splits"]),
                              ko.R(bb["node_coordinates"].astype(dt), bb["node_splits"]),
                              ko.R(bb["edge_indices"], bb["edge_splits"]), depth=3, equiv_method="eps")
             for dt in (np.float32, np.float64)]
    f32, f64 = (tfo.painn_energy_force(P, bb, dt, equiv_method="eps")[1] for dt in (torch.float32, torch.float64))
    eng = np.stack([np.asarray(o["energy"]).reshape(-1) for o in out])
    frc = np.concatenate([np.asarray(o["forces"]) for o in out], axis=0)
    assert_rows_close(eng, e_ref[0], e_ref[1], what=what + " energy")
    assert_forces_close(frc, f32, f64, bb["node_splits"], what=what + " forces")


def _graphs(b, xyz=None):
    ns, es = b["node_splits"], b["edge_splits"]
    xyz = b["node_coordinates"] if xyz is None else xyz
    return [{"node_number": b["node_number"][ns[i]:ns[i + 1]], "node_coordinates": xyz[ns[i]:ns[i + 1]],
             "range_indices": b["edge_indices"][es[i]:es[i + 1]]} for i in range(len(ns) - 1)]


def test_predictor_matches_direct_model_call_and_applies_postprocessors():
    from gcnn_keras_amd.moldyn import MolDynamicsModelPredictor
    from gcnn_keras_amd.ragged import RaggedTensor
    b = synth.md17_like_batch(num_graphs=3, seed=5)
    model = _painn_ef()
    seen = []

    def post(graph, pre_graph):
        seen.append(len(pre_graph["node_number"]))
        return {"energy_ev": graph["energy"] * 27.2114}

    predictor = MolDynamicsModelPredictor(model=model, model_inputs=ITEMS,
                                          model_outputs={"energy": "energy", "forces": "force"},
                                          graph_postprocessors=[post], store_last_input=True)
    out = predictor(_graphs(b))
    assert len(out) == 3 and seen == [21, 21, 21]
    _check_against_oracle(out, b, what="MD predictor, 3 molecules")
    direct = model([RaggedTensor.from_numpy(b["node_number"], b["node_splits"]),
                    RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
                    RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])])
    e = direct["energy"].cpu().numpy()
    f = direct["force"].numpy_rows()
    for i in range(3):
        assert np.array_equal(out[i]["energy"], e[i])
        assert np.array_equal(out[i]["forces"], f[i]) and out[i]["forces"].shape == (21, 3)
        assert np.allclose(out[i]["energy_ev"], e[i] * 27.2114)
    assert predictor._last_input is not None and predictor._counter == 1
    with pytest.raises(TypeError):
        predictor._translate_properties(direct, 5)


def test_graph_replayed_md_steps_equal_eager_and_recapture_on_new_topology():
    from gcnn_keras_amd.moldyn import MolDynamicsModelPredictor
    model = _painn_ef()
    b = synth.md17_like_batch(num_graphs=1, seed=6)      # one molecule, the MD use case
    eager = MolDynamicsModelPredictor(model=model, model_inputs=ITEMS, model_outputs={"energy": "energy", "forces": "force"})
    fast = MolDynamicsModelPredictor(model=model, model_inputs=ITEMS, model_outputs={"energy": "energy", "forces": "force"},
                                     use_graph=True)
    rng = np.random.default_rng(0)
    xyz = b["node_coordinates"].copy()
    for step in range(4):                                 # same neighbour list, moving atoms
        ref = eager(_graphs(b, xyz))
        got = fast(_graphs(b, xyz))
        _check_against_oracle(got, b, xyz, what="MD step %d (graph replay)" % step)
        _check_against_oracle(ref, b, xyz, what="MD step %d (eager)" % step)
        scale = np.max(np.abs(ref[0]["forces"]))
        assert np.max(np.abs(got[0]["energy"] - ref[0]["energy"])) <= 1e-6 * max(1.0, abs(float(ref[0]["energy"][0])))
        assert np.max(np.abs(got[0]["forces"] - ref[0]["forces"])) <= 1e-5 * scale
        xyz = xyz + rng.normal(scale=0.01, size=xyz.shape).astype(np.float32)
    assert fast.graph_captures == 1
    assert fast.fast_steps >= 2                           # same neighbour list: the steps after the capture skip the packer
    b2 = dict(b)
    keep = np.ones(len(b["edge_indices"]), dtype=bool)
    keep[3] = False                                       # one pair leaves the cutoff: new topology -> new graph
    b2["edge_indices"] = b["edge_indices"][keep]
    b2["edge_splits"] = np.array([0, keep.sum()], dtype=np.int64)
    ref = eager(_graphs(b2, xyz))
    got = fast(_graphs(b2, xyz))
    assert fast.graph_captures == 2
    _check_against_oracle(got, b2, xyz, what="MD step on the new topology")
    steps_before = fast.fast_steps
    again = fast(_graphs(b2, xyz + 0.01))                 # ... and the fast step follows the new capture
    assert fast.fast_steps == steps_before + 1
    _check_against_oracle(again, b2, xyz + 0.01, what="MD fast step on the new topology")
    assert np.max(np.abs(got[0]["forces"] - ref[0]["forces"])) <= 1e-5 * np.max(np.abs(ref[0]["forces"]))
    t_eager = eager._test_timing(_graphs(b2, xyz), repetitions=5)
    t_fast = fast._test_timing(_graphs(b2, xyz), repetitions=5)
    print("MD step (PaiNN energy+force, 21 atoms): eager %.2f ms, HIP-graph replay %.2f ms" % (t_eager * 1e3, t_fast * 1e3))
    assert t_fast < t_eager


def test_on_device_set_range_as_tensor_preprocessor():
    from gcnn_keras_amd.graph.preprocessor import SetRange
    from gcnn_keras_amd.moldyn import MolDynamicsModelPredictor
    model = _painn_ef()
    b = synth.md17_like_batch(num_graphs=2, seed=7)      # edges made by the reference rule: max_distance 5, no cap
    with_host_edges = MolDynamicsModelPredictor(model=model, model_inputs=ITEMS,
                                                model_outputs={"energy": "energy", "forces": "force"})
    on_device = MolDynamicsModelPredictor(model=model, model_inputs=ITEMS,
                                          model_outputs={"energy": "energy", "forces": "force"},
                                          tensor_preprocessors=[SetRange(max_distance=5.0, max_neighbours=10000)])
    graphs = _graphs(b)
    ref = with_host_edges(graphs)
    got = on_device([{k: v for k, v in g.items() if k != "range_indices"} for g in graphs])
    _check_against_oracle(got, b, what="MD predictor with on-device SetRange")
    for i in range(2):
        assert np.array_equal(got[i]["energy"], ref[i]["energy"])
        assert np.array_equal(got[i]["forces"], ref[i]["forces"])
