"""EnergyForceModel (kgcnn/model/force.py:136-201): energies against the NumPy oracle, forces - every atom of every
molecule - against the analytic reference oracle/torch_force_oracle.py (torch-CPU autograd restatement of force.py:159-186
in float64, with its float32 twin as the error budget; checked against the NumPy oracle and finite differences in
tests/test_force_oracle.py) through ``parity.assert_forces_close``."""
import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth
from oracle import kgcnn_oracle as ko
from oracle import torch_force_oracle as tfo
from parity import assert_forces_close, assert_rows_close

pytestmark = pytest.mark.gpu


def _dev(values, splits):
    from gcnn_keras_amd.ragged import RaggedTensor
    return RaggedTensor.from_numpy(values, splits)


def test_painn_energy_force_config3_shape():
    from gcnn_keras_amd.literature import PAiNN
    from gcnn_keras_amd.model.force import EnergyForceModel
    b = synth.md17_like_batch(num_graphs=2, seed=5)
    p = synth.painn_params(seed=8, random_bias=True)
    energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
    order = ["embedding", "bessel/frequencies"]
    for i in range(3):
        order += ["conv%d/dense1/kernel" % i, "conv%d/dense1/bias" % i, "conv%d/phi/kernel" % i, "conv%d/phi/bias" % i,
                  "conv%d/w/kernel" % i, "conv%d/w/bias" % i,
                  "update%d/dense1/kernel" % i, "update%d/dense1/bias" % i, "update%d/lin_u/kernel" % i,
                  "update%d/lin_v/kernel" % i, "update%d/a/kernel" % i, "update%d/a/bias" % i]
    order += ["output_mlp/0/kernel", "output_mlp/0/bias", "output_mlp/1/kernel", "output_mlp/1/bias"]
    energy.set_weights([p[k] for k in order])
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_as_dict=True,
                             output_to_tensor=True, output_squeeze_states=True)
    out = model([_dev(b["node_number"], b["node_splits"]), _dev(b["node_coordinates"], b["node_splits"]),
                 _dev(b["edge_indices"], b["edge_splits"])])
    eng, force = out["energy"].cpu().numpy(), out["force"].cpu().numpy()
    assert eng.shape == (2, 1) and force.shape == (2, 21, 3)

    def energy_fn(dtype):
        return ko.painn_forward(ko.to_dtype(p, dtype), ko.R(b["node_number"], b["node_splits"]),
                                ko.R(b["node_coordinates"].astype(dtype), b["node_splits"]),
                                ko.R(b["edge_indices"], b["edge_splits"]), depth=3, equiv_method="eps")

    assert_rows_close(eng, energy_fn(np.float32), energy_fn(np.float64), what="PaiNN energy, 2 graphs")
    f32, f64 = (tfo.painn_energy_force(p, b, dt, equiv_method="eps")[1] for dt in (torch.float32, torch.float64))
    assert_forces_close(force, f32, f64, b["node_splits"], what="PaiNN forces, 2 graphs")


def test_schnet_energy_force_ragged_output_and_tuple_quirk():
    from gcnn_keras_amd.literature import Schnet
    from gcnn_keras_amd.model.force import EnergyForceModel
    b = synth.qm9_like_batch(num_graphs=3, seed=9)
    p = synth.schnet_params(seed=7, random_bias=True)
    energy = Schnet.make_model(depth=3)
    energy.set_weights(list(p.values()))
    # reference quirk: the default energy_output=1 forces output_as_dict=False -> (energy, force) tuple (force.py:115-117)
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, output_to_tensor=False, output_squeeze_states=True)
    assert model.output_as_dict is False
    eng, force = model([_dev(b["node_number"], b["node_splits"]), _dev(b["node_coordinates"], b["node_splits"]),
                        _dev(b["edge_indices"], b["edge_splits"])])
    assert tuple(force.values.shape) == (int(b["node_splits"][-1]), 3)
    got = force.values.cpu().numpy()
    f32, f64 = (tfo.schnet_energy_force(p, b, dt)[1] for dt in (torch.float32, torch.float64))
    assert_forces_close(got, f32, f64, b["node_splits"], what="SchNet forces, 3 graphs")
    # forces of each molecule sum to ~0 (translation invariance): a size-independent property
    for g in range(3):
        blk = got[b["node_splits"][g]:b["node_splits"][g + 1]]
        assert np.max(np.abs(blk.sum(0))) <= 2e-5 * np.max(np.abs(blk))


def test_energy_force_model_config_and_errors():
    from gcnn_keras_amd.model.force import EnergyForceModel
    with pytest.raises(ValueError):
        EnergyForceModel(model_energy=None)
    m = EnergyForceModel(model_energy={"module_name": "kgcnn.literature.Schnet", "class_name": "make_model",
                                       "config": {"depth": 1}}, energy_output=0)
    cfg = m.get_config()
    assert cfg["coordinate_input"] == 1 and cfg["output_as_dict"] is True and cfg["model_energy"]["config"]["depth"] == 1


def test_schnet_energy_force_at_64_graphs():
    """Energy + forces at BASELINE batch size: energy rows against the oracle, every atom's force against the analytic
    reference, per-molecule force sums ~ 0."""
    from gcnn_keras_amd.literature import Schnet
    from gcnn_keras_amd.model.force import EnergyForceModel
    from helpers import mol_inputs
    b = synth.qm9_like_batch(num_graphs=64, seed=2345)
    p = synth.schnet_params(seed=7, random_bias=True)
    energy = Schnet.make_model(depth=3)
    energy.set_weights(list(p.values()))
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False,
                             output_squeeze_states=True)
    out = model(mol_inputs(b))
    eng, force = out["energy"].cpu().numpy(), out["force"].values.cpu().numpy()
    ref = ko.schnet_forward(p, ko.R(b["node_number"], b["node_splits"]), ko.R(b["node_coordinates"], b["node_splits"]),
                            ko.R(b["edge_indices"], b["edge_splits"]), depth=3)
    assert_rows_close(eng, ref, what="SchNet energy at 64 graphs")
    f32, f64 = (tfo.schnet_energy_force(p, b, dt)[1] for dt in (torch.float32, torch.float64))
    assert_forces_close(force, f32, f64, b["node_splits"], what="SchNet forces at 64 graphs")
    ns = b["node_splits"]
    for g in range(64):
        blk = force[ns[g]:ns[g + 1]]
        assert np.max(np.abs(blk.sum(0))) <= 2e-5 * np.max(np.abs(f64[ns[g]:ns[g + 1]])), g


def test_energy_force_model_esp_branch():
    """QM/MM branch (kgcnn/model/force.py:153-158, 165-168, 179-186): the energy model consumes the electrostatic
    potential at the atoms, esp (batch,[N]); the force gains dE/desp * desp/dr.  The energy model here is SchNet on 2-D
    node attributes [features | esp]; reference forces: the analytic float64 / float32 restatement of that formula
    (oracle/torch_force_oracle.energy_force with esp, desp_dr)."""
    from gcnn_keras_amd.layers.modules import concat_last
    from gcnn_keras_amd.literature import Schnet
    from gcnn_keras_amd.model.force import EnergyForceModel
    from helpers import dev
    b = synth.qm9_like_batch(num_graphs=2, seed=19)
    n = int(b["node_splits"][-1])
    rng = np.random.default_rng(5)
    feat = rng.normal(size=(n, 4)).astype(np.float32)
    esp = rng.normal(scale=0.3, size=(n,)).astype(np.float32)
    desp_dr = rng.normal(scale=0.2, size=(n, 3)).astype(np.float32)
    p = synth.schnet_params(seed=7, random_bias=True, emb_out=5)
    del p["embedding"]                                    # 2-D node attributes: no embedding layer
    schnet = Schnet.make_model(depth=2, inputs=[{"shape": (None, 5), "name": "node_attributes", "dtype": "float32",
                                                 "ragged": True},
                                                {"shape": (None, 3), "name": "node_coordinates", "dtype": "float32",
                                                 "ragged": True},
                                                {"shape": (None, 2), "name": "edge_indices", "dtype": "int64",
                                                 "ragged": True}])
    keep = [k for k in p if not k.startswith("interaction2/")]
    schnet.set_weights([p[k] for k in keep])
    assert schnet.fused is None                            # attributes instead of numbers: layer path

    def energy_model(inputs, **kwargs):
        f, xyz, idx, e, _ = inputs
        attr = f.with_values(concat_last([f.values, e.values.unsqueeze(-1)]))
        return schnet([attr, xyz, idx])

    model = EnergyForceModel(model_energy=energy_model, coordinate_input=1, esp_input=3, esp_grad_input=4,
                             energy_output=0, output_to_tensor=False, output_squeeze_states=True)
    assert model.get_config()["esp_input"] == 3 and model.get_config()["esp_grad_input"] == 4
    inputs = [dev(feat, b["node_splits"]), dev(b["node_coordinates"], b["node_splits"]),
              dev(b["edge_indices"], b["edge_splits"]), dev(esp, b["node_splits"]), dev(desp_dr, b["node_splits"])]
    out = model(inputs)
    eng, force = out["energy"].cpu().numpy(), out["force"].values.cpu().numpy()
    pk = {k: p[k] for k in keep}

    def oracle_energy(dtype):
        attr = np.concatenate([feat.astype(dtype), esp.astype(dtype)[:, None]], axis=1)
        return ko.schnet_forward(ko.to_dtype(pk, dtype), ko.R(attr, b["node_splits"]),
                                 ko.R(b["node_coordinates"].astype(dtype), b["node_splits"]),
                                 ko.R(b["edge_indices"], b["edge_splits"]), depth=2)

    assert_rows_close(eng, oracle_energy(np.float32), oracle_energy(np.float64), what="ESP energy")

    def reference(dtype, with_chain=True):
        pt = tfo.to_torch(pk, dtype)
        ft = torch.from_numpy(feat).to(dtype)
        fn = lambda x, e: tfo.schnet_energy(pt, torch.cat([ft, e.unsqueeze(-1)], dim=1), x, b["edge_indices"],
                                            b["node_splits"], b["edge_splits"], depth=2)
        return tfo.energy_force(fn, b["node_coordinates"], dtype, esp=esp, desp_dr=desp_dr if with_chain else None)[1][..., 0]

    f32, f64 = reference(torch.float32), reference(torch.float64)
    plain64 = reference(torch.float64, with_chain=False)
    assert np.max(np.abs(f64 - plain64)) > 0.05 * np.max(np.abs(f64))            # the chain term matters here
    assert_forces_close(force, f32, f64, b["node_splits"], what="ESP branch forces")
    # with only one of the two inputs named the branch is not taken (force.py:153): plain -dE/dx
    plain = EnergyForceModel(model_energy=energy_model, coordinate_input=1, esp_input=3, energy_output=0,
                             output_to_tensor=False, output_squeeze_states=True)(inputs)
    assert_forces_close(plain["force"].values.cpu().numpy(), reference(torch.float32, False), plain64, b["node_splits"],
                        what="ESP inputs incomplete: plain forces")


FORK_SCHNET = dict(
    inputs=[{"shape": [None], "name": "node_number", "dtype": "int64", "ragged": True},
            {"shape": [None, 3], "name": "node_coordinates", "dtype": "float32", "ragged": True},
            {"shape": [None, 2], "name": "range_indices", "dtype": "int64", "ragged": True}],
    input_embedding={"node": {"input_dim": 95, "output_dim": 128}},
    interaction_args={"units": 128, "use_bias": True, "activation": "shifted_softplus", "cfconv_pool": "sum"},
    node_pooling_args={"pooling_method": "sum"}, depth=6,
    gauss_args={"bins": 25, "distance": 5, "offset": 0.0, "sigma": 0.4}, verbose=10,
    last_mlp={"use_bias": [True] * 3, "units": [128, 64, 1], "activation": ["shifted_softplus"] * 2 + ["linear"]},
    output_embedding="graph", output_to_tensor=True, use_output_mlp=False, output_mlp=None)


def _fork_schnet_case(num_graphs, seed):
    """The fork's force_schnet.py model (force_schnet.py:33-45, 128-156): int64 node numbers, embedding 128, depth 6,
    Gauss(25, 5.0, 0.4), last_mlp [128, 64, 1] ending linear, no output MLP."""
    from gcnn_keras_amd.literature import Schnet
    from helpers import dev
    b = synth.md17_like_batch(num_graphs=num_graphs, seed=seed)
    p = synth.schnet_params(seed=7, depth=6, emb_out=128, bins=25, last_units=(128, 64, 1), out_units=(),
                            random_bias=True)
    model = Schnet.make_model(**FORK_SCHNET)
    model.set_weights(list(p.values()))
    inputs = [dev(b["node_number"].astype(np.int64), b["node_splits"]), dev(b["node_coordinates"], b["node_splits"]),
              dev(b["edge_indices"], b["edge_splits"])]
    oracle = lambda xyz, pp: ko.schnet_forward(
        pp, ko.R(b["node_number"], b["node_splits"]), ko.R(xyz, b["node_splits"]), ko.R(b["edge_indices"], b["edge_splits"]),
        depth=6, gauss_args=FORK_SCHNET["gauss_args"], last_mlp_act=("kgcnn>shifted_softplus",) * 2 + ("linear",),
        output_mlp_act=())
    return b, p, model, inputs, oracle


def test_fork_schnet_configuration_takes_the_fused_forward():
    from parity import assert_rows_close
    b, p, model, inputs, oracle = _fork_schnet_case(16, 5)
    assert model.fused is not None and model.fused.accepts(inputs)
    out = model(inputs)
    assert model.fused.last == "direct" and torch.equal(model(inputs), out) and model.fused.last == "graph"
    got = out.cpu().numpy()
    assert got.shape == (16, 1)
    assert_rows_close(got, oracle(b["node_coordinates"], p), oracle(b["node_coordinates"].astype(np.float64),
                                                                     ko.to_dtype(p, np.float64)), what="fork SchNet")
    layers = model(inputs, fused=False).cpu().numpy()
    assert_rows_close(layers, oracle(b["node_coordinates"], p), what="fork SchNet, layer path")


@pytest.mark.parametrize("fork", [True, False])
def test_schnet_fused_energy_force(fork):
    """Energy + forces of a SchNet energy model from one HIP graph (fused forward, hand-written reverse pass: the cfconv
    kernel with swapped index columns for dE/dx_j, the distance-gradient MFMA kernel for dE/dd) - for the fork's
    force_schnet.py configuration and for the reference default head - against the oracle energy, float64 finite
    differences per molecule, the tape + layer path, and the zero-net-force property at 64 graphs."""
    from gcnn_keras_amd.model.force import EnergyForceModel
    from helpers import mol_inputs
    if fork:
        b, p, energy, inputs, oracle = _fork_schnet_case(64, 2345)
        depth, kw = 6, dict(gauss_args=FORK_SCHNET["gauss_args"],
                            last_mlp_act=("kgcnn>shifted_softplus",) * 2 + ("linear",), output_mlp_act=())
    else:
        from gcnn_keras_amd.literature import Schnet
        b = synth.qm9_like_batch(num_graphs=64, seed=2345)
        p = synth.schnet_params(seed=7, random_bias=True)
        energy = Schnet.make_model(depth=3)
        energy.set_weights(list(p.values()))
        inputs, depth, kw = mol_inputs(b), 3, {}
        oracle = lambda xyz, pp: ko.schnet_forward(pp, ko.R(b["node_number"], b["node_splits"]), ko.R(xyz, b["node_splits"]),
                                                   ko.R(b["edge_indices"], b["edge_splits"]), depth=3)
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False,
                             output_squeeze_states=True, is_physical_force=not fork, output_as_dict=not fork)
    first = model(inputs)
    assert energy.fused.last == "eager"
    second = model(inputs)
    assert energy.fused.last == "graph"
    eng, frc = (first["energy"], first["force"]) if not fork else first      # the fork asks for a tuple of outputs
    eng2, frc2 = (second["energy"], second["force"]) if not fork else second
    assert torch.equal(eng, eng2) and torch.equal(frc.values, frc2.values)
    eng, force = eng.cpu().numpy(), frc.values.cpu().numpy()
    if fork:
        force = -force                                     # is_physical_force=False: the model returned +dE/dx
    g_count = len(b["node_splits"]) - 1
    assert eng.shape == (g_count, 1) and force.shape == (int(b["node_splits"][-1]), 3)
    assert_rows_close(eng, oracle(b["node_coordinates"], p), what="SchNet energy (fused force pass)")
    f32, f64 = (tfo.schnet_energy_force(p, b, dt, depth=depth, **kw)[1] for dt in (torch.float32, torch.float64))
    assert_forces_close(force, f32, f64, b["node_splits"], what="fused SchNet forces (fork=%s), 64 graphs" % fork)
    ns = b["node_splits"]
    for g in range(g_count):
        assert np.max(np.abs(force[ns[g]:ns[g + 1]].sum(0))) <= 2e-5 * np.max(np.abs(f64[ns[g]:ns[g + 1]])), g
    model.fused = False                                    # tape + layer-by-layer reverse pass: same reference, same bar
    ref = model(inputs)
    ref_f = (ref["force"] if not fork else ref[1]).values.cpu().numpy() * (-1.0 if fork else 1.0)
    assert_forces_close(ref_f, f32, f64, b["node_splits"], what="tape SchNet forces (fork=%s), 64 graphs" % fork)


def test_schnet_fused_energy_force_unsorted_edges_weight_update_and_empty_graphs():
    """The force route on a batch whose receivers are shuffled inside every graph (both CSR permutations in use), with a
    trailing graph without edges... then after an in-place weight update (images of the reverse pass re-packed, same
    captured graph): fused and tape routes both against the oracle's energies and analytic forces for the weights in use."""
    from gcnn_keras_amd.literature import Schnet
    from gcnn_keras_amd.model.force import EnergyForceModel
    from helpers import dev
    b = synth.qm9_like_batch(num_graphs=12, seed=77)
    rng = np.random.default_rng(1)
    idx, es = b["edge_indices"].copy(), b["edge_splits"]
    for g in range(len(es) - 1):
        idx[es[g]:es[g + 1]] = idx[es[g]:es[g + 1]][rng.permutation(es[g + 1] - es[g])]
    p = synth.schnet_params(seed=3, random_bias=True)
    energy = Schnet.make_model(depth=3)
    energy.set_weights(list(p.values()))
    inputs = [dev(b["node_number"], b["node_splits"]), dev(b["node_coordinates"], b["node_splits"]), dev(idx, es)]
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False,
                             output_squeeze_states=True)

    bb = dict(b, edge_indices=idx)

    def both():
        pw = dict(zip(p.keys(), energy.get_weights()))      # the weights the model holds now (constructor order)
        e_ref = [ko.schnet_forward(ko.to_dtype(pw, dt), ko.R(b["node_number"], b["node_splits"]),
                                   ko.R(b["node_coordinates"].astype(dt), b["node_splits"]), ko.R(idx, es), depth=3)
                 for dt in (np.float32, np.float64)]
        f32, f64 = (tfo.schnet_energy_force(pw, bb, dt)[1] for dt in (torch.float32, torch.float64))
        model.fused = None
        model(inputs)
        out = model(inputs)
        assert energy.fused.last == "graph"
        model.fused = False
        ref = model(inputs)
        for name, o in (("fused", out), ("tape", ref)):
            # after the weight update one molecule's energy is -1.5e-3 next to 2.8 for its neighbour (atom terms cancel):
            # measured against the 1e-3 floor the float32 oracle itself is 8e-5 from float64 on that row, hence the cap
            assert_rows_close(o["energy"].cpu().numpy(), e_ref[0], e_ref[1], cap=2e-4,
                              what="SchNet energy, unsorted edges (%s)" % name)
            assert_forces_close(o["force"].values.cpu().numpy(), f32, f64, b["node_splits"],
                                what="SchNet forces, unsorted edges (%s)" % name)
        return out["force"].values.cpu().numpy()

    f0 = both()
    slots = len(energy.fused._gslots)
    with torch.no_grad():
        for name, t in energy.weights:
            if name.endswith("kernel"):
                t.mul_(1.05)
    f1 = both()
    assert len(energy.fused._gslots) == slots and np.max(np.abs(f1 - f0)) > 1e-3 * np.max(np.abs(f0))


@pytest.mark.parametrize("family", ["schnet", "painn"])
def test_fused_force_routes_honour_sign_states_axis_and_call_kwargs(family):
    """``is_physical_force=False`` (+dE/dx, force.py:185-186) and ``output_squeeze_states=False`` (forces (N, 3, 1),
    force.py:187-188) on the fused energy + force routes, against the analytic reference; ``model(inputs, fused=False)``
    and ``training=True`` reach the energy model and take the tape instead of the fused reverse pass."""
    from gcnn_keras_amd.model.force import EnergyForceModel
    from helpers import mol_inputs, painn_weight_list
    if family == "schnet":
        from gcnn_keras_amd.literature import Schnet
        b = synth.qm9_like_batch(num_graphs=6, seed=41)
        p = synth.schnet_params(seed=7, random_bias=True)
        energy = Schnet.make_model(depth=3)
        energy.set_weights(list(p.values()))
        ref = lambda dt: tfo.schnet_energy_force(p, b, dt, is_physical_force=False)[1]
    else:
        from gcnn_keras_amd.literature import PAiNN
        b = synth.md17_like_batch(num_graphs=4, seed=42)
        p = synth.painn_params(seed=8, random_bias=True)
        energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
        energy.set_weights(painn_weight_list(p))
        ref = lambda dt: tfo.painn_energy_force(p, b, dt, is_physical_force=False, equiv_method="eps")[1]
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False,
                             output_squeeze_states=False, is_physical_force=False)
    x = mol_inputs(b)
    out = model(x)
    assert energy.fused.last in ("eager", "direct")          # the fused route ran
    grad = out["force"].values.cpu().numpy()
    assert grad.shape == (int(b["node_splits"][-1]), 3, 1)
    g32, g64 = ref(torch.float32), ref(torch.float64)
    assert_forces_close(grad[..., 0], g32, g64, b["node_splits"], what="%s +dE/dx, states axis kept" % family)
    calls_before = energy.fused.last
    energy.fused.last = None
    tape = model(x, fused=False)                              # the kwarg reaches the energy model: layer path + tape
    assert energy.fused.last is None
    assert_forces_close(tape["force"].values.cpu().numpy()[..., 0], g32, g64, b["node_splits"],
                        what="%s +dE/dx through the tape" % family)
    del calls_before
