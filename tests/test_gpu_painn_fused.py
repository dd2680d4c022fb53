"""Fused PaiNN pipeline (gcnn_keras_amd/fused_painn.py, csrc/mp_painn_fused.hip) behind ``PAiNN.make_model`` and behind
``EnergyForceModel``: BASELINE config 3 (64 MD17-shaped graphs, energy + forces) against the CPU oracle, against the
layer path, and through size-independent properties (forces of a molecule sum to zero; graphs are independent).

Forces: every atom of every molecule against the analytic reference oracle/torch_force_oracle.py (torch-CPU autograd
restatement of kgcnn/model/force.py:159-186, float64 truth + float32 twin; itself checked against the NumPy oracle's
energies and finite differences in tests/test_force_oracle.py) through ``parity.assert_forces_close``."""
import numpy as np
import pytest
import torch

from gcnn_keras_amd import synth
from helpers import mol_inputs, painn_weight_list
from oracle import kgcnn_oracle as ko
from oracle import torch_force_oracle as tfo
from parity import assert_forces_close, assert_rows_close, rowwise_rel

pytestmark = pytest.mark.gpu


def _model(p, **kw):
    from gcnn_keras_amd.literature import PAiNN
    kw.setdefault("equiv_initialize_kwargs", {"dim": 3, "method": "eps"})
    model = PAiNN.make_model(**kw)
    model.set_weights(painn_weight_list(p, depth=kw.get("depth", 3)))
    return model


def _oracle(p, b, dtype=np.float32, cutoff=None, depth=3, xyz=None):
    pp = ko.to_dtype(p, dtype)
    xyz = b["node_coordinates"] if xyz is None else xyz
    return ko.painn_forward(pp, ko.R(b["node_number"], b["node_splits"]), ko.R(xyz.astype(dtype), b["node_splits"]),
                            ko.R(b["edge_indices"], b["edge_splits"]), depth=depth, equiv_method="eps", cutoff=cutoff)


def _reference_forces(p, b, cutoff=None):
    """(float32, float64) analytic forces (N, 3) of the whole batch."""
    return tuple(tfo.painn_energy_force(p, b, dt, equiv_method="eps", cutoff=cutoff)[1]
                 for dt in (torch.float32, torch.float64))


@pytest.mark.parametrize("num_graphs,seed", [(2, 5), (64, 2345)])
def test_painn_make_model_fused_forward(num_graphs, seed):
    from gcnn_keras_amd import _ffi
    b = synth.md17_like_batch(num_graphs=num_graphs, seed=seed)
    p = synth.painn_params(seed=8, random_bias=True)
    model = _model(p)
    assert model.fused is not None
    x = mol_inputs(b)
    model.fused._sync_weights()          # weight images (one pack launch per matrix) are made once per weight update
    before = _ffi.launch_count()
    out1 = model(x)
    # stage 0, per block 3 chain launches + 3 memory-bound kernels, readout (pool + MLP), + index plan at bind
    assert model.fused.last == "eager" and _ffi.launch_count() - before <= 1 + 6 * 3 + 3 + 4
    out2 = model(x)
    assert model.fused.last == "graph" and torch.equal(out1, out2)
    got = out1.cpu().numpy()
    assert got.shape == (num_graphs, 1)
    assert_rows_close(got, _oracle(p, b), _oracle(p, b, np.float64), what="fused PaiNN forward")
    layers = model(x, fused=False).cpu().numpy()
    assert rowwise_rel(layers, got) <= 1e-5
    model.fused.check_flags()


def test_painn_energy_force_config3_fused():
    """BASELINE config 3: 64 aspirin-shaped molecules, energy (64,1) and forces (64,21,3) from ONE captured graph."""
    from gcnn_keras_amd.model.force import EnergyForceModel
    b = synth.md17_like_batch(num_graphs=64, seed=2345)
    assert int(b["node_splits"][-1]) == 1344 and int(b["edge_splits"][-1]) == 20586      # BASELINE.md calibration
    p = synth.painn_params(seed=8, random_bias=True)
    energy = _model(p)
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_as_dict=True,
                             output_to_tensor=True, output_squeeze_states=True)
    x = mol_inputs(b)
    out = model(x)
    assert energy.fused.last == "eager"
    out2 = model(x)
    assert energy.fused.last == "graph"
    eng, force = out["energy"].cpu().numpy(), out["force"].cpu().numpy()
    assert eng.shape == (64, 1) and force.shape == (64, 21, 3)
    assert np.array_equal(eng, out2["energy"].cpu().numpy()) and np.array_equal(force, out2["force"].cpu().numpy())
    assert_rows_close(eng, _oracle(p, b), _oracle(p, b, np.float64), what="fused PaiNN energy")
    # every atom of all 64 molecules against the analytic float64 forces
    f32, f64 = _reference_forces(p, b)
    assert_forces_close(force, f32, f64, b["node_splits"], what="fused PaiNN forces, config 3")
    # size-independent property: the forces of every molecule sum to zero (translation invariance of the energy)
    mol_scale = np.max(np.abs(f64.reshape(64, 21, 3)), axis=(1, 2))
    assert np.max(np.abs(force.sum(axis=1)) / mol_scale[:, None]) <= 2e-5
    # the tape + layer-by-layer reverse pass is held to the same reference
    model.fused = False
    ref_layers = model(x)
    assert_forces_close(ref_layers["force"].cpu().numpy(), f32, f64, b["node_splits"], what="tape PaiNN forces, config 3")
    assert rowwise_rel(ref_layers["energy"].cpu().numpy(), eng) <= 1e-5


def test_painn_fused_cutoff_envelope_unsorted_edges_and_weight_update():
    """conv cutoff (cosine envelope on the filter, incl. its derivative in the reverse pass), receivers shuffled inside
    the graphs (stable-sort route), trailing graph without edges, and a weight update after the batch was captured."""
    from gcnn_keras_amd.model.force import EnergyForceModel
    b = synth.md17_like_batch(num_graphs=3, seed=77)
    rng = np.random.default_rng(3)
    idx = b["edge_indices"].copy()
    for g in range(3):
        lo, hi = b["edge_splits"][g], b["edge_splits"][g + 1]
        idx[lo:hi] = idx[lo:hi][rng.permutation(hi - lo)]
    b["edge_indices"] = idx
    # a fourth molecule with two atoms further apart than the 5 A edge cutoff: no edges
    b["node_number"] = np.concatenate([b["node_number"], np.array([6., 8.], np.float32)])
    b["node_coordinates"] = np.concatenate([b["node_coordinates"], np.array([[0, 0, 0], [9, 0, 0]], np.float32)])
    b["node_splits"] = np.concatenate([b["node_splits"], [b["node_splits"][-1] + 2]])
    b["edge_splits"] = np.concatenate([b["edge_splits"], [b["edge_splits"][-1]]])
    conv = {"units": 128, "cutoff": 5.0, "conv_pool": "sum"}
    for seed in (8, 9):
        p = synth.painn_params(seed=seed, random_bias=True)
        if seed == 8:
            energy = _model(p, conv_args=conv)
            model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False,
                                     output_squeeze_states=True)
            x = mol_inputs(b)
            model(x), model(x)                       # bound and captured with the first weights
        else:
            energy.set_weights(painn_weight_list(p))  # same tensors, new values: derived layouts are refreshed
        out = model(x)
        eng, force = out["energy"].cpu().numpy(), out["force"].values.cpu().numpy()
        assert eng.shape == (4, 1) and force.shape == (65, 3)
        # four energies, one of them 10x smaller than the largest: float32 pipelines (oracle, layer path, fused) all sit
        # 1e-6 of the LARGEST energy from float64 here (scripts/diag_painn_cutoff.py), which is 1e-5 of the small row -
        # rows are therefore measured against 10 % of the scale
        assert_rows_close(eng, _oracle(p, b, cutoff=5.0), _oracle(p, b, np.float64, cutoff=5.0), floor=0.1,
                          what="cutoff energy")
        f32, f64 = _reference_forces(p, b, cutoff=5.0)
        assert np.all(f64[-2:] == 0.0)               # the edgeless molecule feels no force
        assert_forces_close(force, f32, f64, b["node_splits"], what="PaiNN forces, cutoff envelope, seed %d" % seed)
    assert energy.fused.last == "graph"


def test_painn_route_falls_back_when_it_must():
    from gcnn_keras_amd.literature import PAiNN
    assert PAiNN.make_model(conv_args={"units": 64, "cutoff": None, "conv_pool": "sum"}, update_args={"units": 64},
                            input_embedding={"node": {"input_dim": 95, "output_dim": 64}}).fused is None
    assert PAiNN.make_model(pooling_args={"pooling_method": "mean"}).fused is None
    assert PAiNN.make_model(output_embedding="node").fused is None
    two = PAiNN.make_model(output_mlp={"use_bias": [True, True], "units": [128, 2], "activation": ["swish", "linear"]})
    assert two.fused is not None and not two.fused.single_state     # forward fused, forces take the tape (two states)
    b = synth.md17_like_batch(num_graphs=2, seed=1)
    from gcnn_keras_amd.model.force import EnergyForceModel
    out = EnergyForceModel(model_energy=two, energy_output=0, output_to_tensor=False)(mol_inputs(b))
    assert tuple(out["energy"].shape) == (2, 2) and tuple(out["force"].values.shape) == (42, 3, 2)


def test_message_tiles_in_lds_against_gather_route_and_limits(monkeypatch):
    """The LDS-tile message kernel (``mp_painn_message_tiles_f32``: node tiles staged by LDS-DMA, filter on the matrix
    pipe) is what a receiver-sorted batch runs; the gather kernels (``mp_painn_message_f32``) serve unsorted edge lists and
    ``MPENGINE_PAINN_TILES=0``.  Both routes against the oracle and against each other on QM9-shaped molecules (3-29 atoms:
    tiles of one to four receivers, receivers without edges), repeatedly (run-to-run bit equality); a tile that cannot fit
    LDS is refused with ValueError, never launched."""
    from gcnn_keras_amd import _ffi
    b = synth.qm9_like_batch(num_graphs=40, seed=91, max_distance=5.0, max_neighbours=10000)
    b["node_coordinates"][int(b["node_splits"][7])] += 50.0          # an atom out of everyone's reach: receiver without edges
    b = dict(b, **{k: v for k, v in _edges_by_rule(b).items()})
    p = synth.painn_params(seed=8, random_bias=True)
    ref32, ref64 = _oracle(p, b), _oracle(p, b, np.float64)
    model = _model(p)
    x = mol_inputs(b)
    first = model(x)
    slot = model.fused.slot_of(x)
    assert slot.tiles0 is not None and slot.tiles0["count"] >= 40 and slot.tiles0["max_rows"] <= 29
    for _ in range(5):
        assert torch.equal(model(x), first)
    assert_rows_close(first.cpu().numpy(), ref32, ref64, what="PaiNN forward, LDS tiles")
    monkeypatch.setenv("MPENGINE_PAINN_TILES", "0")
    gather_model = _model(p)
    x2 = mol_inputs(b)
    got = gather_model(x2)
    assert gather_model.fused.slot_of(x2).tiles0 is None
    assert_rows_close(got.cpu().numpy(), ref32, ref64, what="PaiNN forward, gather kernels")
    assert rowwise_rel(got.cpu().numpy(), first.cpu().numpy()) <= 2e-5     # two float32 pipelines, each within 1e-5 of the oracle
    monkeypatch.delenv("MPENGINE_PAINN_TILES")
    blk = slot.blk[0]
    with pytest.raises(ValueError):
        _ffi.call("mp_painn_message_tiles_f32", _ffi.ptr(blk["s"]), _ffi.ptr(slot.v0), slot.N, _ffi.ptr(slot.rbf), slot.B,
                  None, _ffi.ptr(slot.rij), _ffi.ptr(slot.w["conv0/w/F"]), _ffi.ptr(slot.ptr0), _ffi.ptr(slot.send), slot.M,
                  _ffi.ptr(slot.tiles0["table"]), slot.tiles0["count"], 60, 100, None, _ffi.ptr(blk["zp"]),
                  _ffi.ptr(blk["vp"]), _ffi.stream())


def _edges_by_rule(b, max_distance=5.0):
    es = []
    ns = b["node_splits"]
    for g in range(len(ns) - 1):
        es.append(synth.radius_graph(b["node_coordinates"][ns[g]:ns[g + 1]], max_distance, 10000))
    return {"edge_indices": np.concatenate(es).astype(np.int64).reshape(-1, 2),
            "edge_splits": np.concatenate([[0], np.cumsum([len(e) for e in es])]).astype(np.int64)}


def test_painn_launch_group_energy_and_forces():
    """``route.call_group([...], with_forces=True)``: three independent batches (different sizes) concatenated on the device
    and served by one launch sequence - energies and every atom's force of every member against the oracle / the analytic
    reference, equal (2e-6 / same force bars) to calls of their own, replayed bit-identically, following an in-place
    coordinate update of one member.  (Member seeds: batches whose float32 oracle is itself within 1e-5 of its float64 twin -
    a molecule whose atom contributions cancel carries float32 noise of several 1e-5 in all three float32 pipelines, engine,
    layer path and oracle alike: scripts/diag_painn_energy_noise.py, 30 seeds, medians 7.5e-7 / 9.4e-7 / 8.4e-7.)"""
    p = synth.painn_params(seed=8, random_bias=True)
    energy = _model(p)
    batches = [synth.md17_like_batch(num_graphs=g, seed=s) for g, s in ((3, 21), (5, 22), (2, 23))]
    ins = [mol_inputs(b) for b in batches]
    alone = [energy.fused.energy_force(x) for x in ins]
    got = energy.fused.call_group(ins, with_forces=True)
    assert energy.fused.last == "eager" and len(got) == 3
    again = energy.fused.call_group(ins, with_forces=True)
    assert energy.fused.last == "graph"
    for k, b in enumerate(batches):
        (e, f), (e2, f2), (ea, fa) = got[k], again[k], alone[k]
        assert torch.equal(e, e2) and torch.equal(f, f2)
        assert tuple(e.shape) == (len(b["node_splits"]) - 1, 1) and tuple(f.shape) == (int(b["node_splits"][-1]), 3)
        assert rowwise_rel(e.cpu().numpy(), ea.cpu().numpy()) <= 2e-6
        assert_rows_close(e.cpu().numpy(), _oracle(p, b), _oracle(p, b, np.float64), what="PaiNN group energy, member %d" % k)
        f32, f64 = _reference_forces(p, b)
        assert_forces_close(f.cpu().numpy(), f32, f64, b["node_splits"], what="PaiNN group forces, member %d" % k)
    fwd_only = energy.fused.call_group(ins)
    for k in range(3):
        assert rowwise_rel(fwd_only[k].cpu().numpy(), got[k][0].cpu().numpy()) <= 2e-6
    ins[1][1].values.add_(0.01)                                        # member 1 moves; the graph re-concatenates
    batches[1]["node_coordinates"] = batches[1]["node_coordinates"] + np.float32(0.01)
    moved = energy.fused.call_group(ins, with_forces=True)
    assert torch.equal(moved[0][1], got[0][1]) and not torch.equal(moved[1][1], got[1][1])
    f32, f64 = _reference_forces(p, batches[1])
    assert_forces_close(moved[1][1].cpu().numpy(), f32, f64, batches[1]["node_splits"], what="PaiNN group forces after a move")
    energy.fused.check_flags()


@pytest.mark.parametrize("cutoff,shuffle", [(None, False), (5.0, False), (None, True)])
def test_reverse_message_tiles_against_gather_route_and_replays(monkeypatch, cutoff, shuffle):
    """The reverse message step on sender tiles (``mp_painn_message_bwd_tiles_f32``: upstream gradients staged by LDS-DMA,
    basis rows gathered through perm1 and split once per tile, filter and its distance derivative on the matrix pipe) is
    what every bound batch runs for forces - receiver-sorted or not; ``MPENGINE_PAINN_BWD_TILES=0`` keeps the sender-parallel
    VALU kernel.  Both routes against the analytic float64 forces and against each other on QM9-shaped molecules (3-29
    atoms: tiles of one to three senders, steps of 1 to 16 edges, an atom without edges), with and without the cosine
    cutoff envelope, with the edge list shuffled inside every graph; 150 replays bit-identical (the kernel's LDS reads sit
    behind full waits and a drained MFMA burst: csrc/mp_painn_fused.hip, `mfma_drained`); a tile that cannot fit LDS is
    refused with ValueError, never launched."""
    from gcnn_keras_amd import _ffi
    b = synth.qm9_like_batch(num_graphs=24, seed=92, max_distance=5.0, max_neighbours=10000)
    b["node_coordinates"][int(b["node_splits"][5])] += 50.0          # an atom out of everyone's reach: no edges at all
    b = dict(b, **{k: v for k, v in _edges_by_rule(b).items()})
    if shuffle:
        rng = np.random.default_rng(4)
        idx = b["edge_indices"].copy()
        for g in range(len(b["edge_splits"]) - 1):
            lo, hi = int(b["edge_splits"][g]), int(b["edge_splits"][g + 1])
            idx[lo:hi] = idx[lo:hi][rng.permutation(hi - lo)]
        b = dict(b, edge_indices=idx)
    p = synth.painn_params(seed=8, random_bias=True)
    kw = {"conv_args": {"units": 128, "cutoff": cutoff, "conv_pool": "sum"}}
    f32, f64 = _reference_forces(p, b, cutoff=cutoff)
    energy = _model(p, **kw)
    x = mol_inputs(b)
    eng, force = energy.fused.energy_force(x)
    slot = energy.fused.slot_of(x, grad=True)
    assert slot.tiles1 is not None and slot.tiles1["max_own"] <= 3 and slot.tiles1["max_rows"] <= 29
    assert (slot.perm0 is not None) == shuffle
    first = force.clone()
    # cap 2e-4: these random-geometry molecules include ill-conditioned ones on which the float32 autograd reference itself
    # is 4e-5 of the molecule's scale from float64 - the bar stays twice that reference's own distance, per molecule
    assert_forces_close(force.cpu().numpy(), f32, f64, b["node_splits"], what="PaiNN forces, sender tiles", cap=2e-4)
    for _ in range(150):
        e2, f2 = energy.fused.energy_force(x)
        assert torch.equal(f2, first) and torch.equal(e2, eng)
    monkeypatch.setenv("MPENGINE_PAINN_BWD_TILES", "0")
    other = _model(p, **kw)
    x2 = mol_inputs(b)
    eng_v, force_v = other.fused.energy_force(x2)
    assert other.fused.slot_of(x2, grad=True).tiles1 is None
    assert_forces_close(force_v.cpu().numpy(), f32, f64, b["node_splits"], what="PaiNN forces, sender-parallel kernel", cap=2e-4)
    assert_forces_close(force.cpu().numpy(), force_v.cpu().numpy(), f64, b["node_splits"], what="PaiNN forces, tiles vs VALU",
                        cap=2e-4)
    monkeypatch.delenv("MPENGINE_PAINN_BWD_TILES")
    tl, blk = slot.tiles1, slot.blk[0]
    with pytest.raises(ValueError):
        _ffi.call("mp_painn_message_bwd_tiles_f32", _ffi.ptr(blk["s"]), _ffi.ptr(slot.v0), slot.N, _ffi.ptr(slot.rbf),
                  _ffi.ptr(slot.rbfd), slot.B, _ffi.ptr(slot.env), _ffi.ptr(slot.envd), _ffi.ptr(slot.rij),
                  _ffi.ptr(slot.w["conv0/w/F"]), _ffi.ptr(slot.ptr1), _ffi.ptr(slot.perm1), _ffi.ptr(slot.recv), slot.M,
                  _ffi.ptr(tl["table"]), tl["count"], 60, 40, 400, _ffi.ptr(slot.g_zp), _ffi.ptr(slot.g_vp), _ffi.ptr(slot.g_s),
                  None, _ffi.ptr(slot.g_d), _ffi.ptr(slot.g_rij), 0, _ffi.stream())


@pytest.mark.parametrize("num_radial", [10, 12, 31])
def test_tile_kernels_with_other_basis_sizes(num_radial):
    """The generic-B builds of both tile kernels (the 20-function basis has builds of its own): 10 (rows not a multiple of
    16 B: dword LDS-DMA in the forward kernel, two k groups of operand pieces), 12 (16-B pieces, two k groups) and 31 (the
    largest basis with a free bias slot: four k groups) - energies against the NumPy oracle, every atom's force against the
    analytic float64 reference, replays bit-identical."""
    b = synth.md17_like_batch(num_graphs=6, seed=31)
    p = synth.painn_params(seed=8, random_bias=True, num_radial=num_radial)
    ba = {"num_radial": num_radial, "cutoff": 5.0, "envelope_exponent": 5}
    energy = _model(p, bessel_basis=ba)
    x = mol_inputs(b)
    eng, force = energy.fused.energy_force(x)
    slot = energy.fused.slot_of(x, grad=True)
    assert slot.B == num_radial and slot.tiles0 is not None and slot.tiles1 is not None
    pp32, pp64 = ko.to_dtype(p, np.float32), ko.to_dtype(p, np.float64)
    ref = [ko.painn_forward(pp, ko.R(b["node_number"], b["node_splits"]), ko.R(b["node_coordinates"].astype(dt), b["node_splits"]),
                            ko.R(b["edge_indices"], b["edge_splits"]), depth=3, equiv_method="eps", bessel_args=ba)
           for pp, dt in ((pp32, np.float32), (pp64, np.float64))]
    assert_rows_close(eng.cpu().numpy(), ref[0], ref[1], what="PaiNN energy, %d basis functions" % num_radial)
    f32, f64 = (tfo.painn_energy_force(p, b, dt, equiv_method="eps", bessel_args=ba)[1] for dt in (torch.float32, torch.float64))
    assert_forces_close(force.cpu().numpy(), f32, f64, b["node_splits"], what="PaiNN forces, %d basis functions" % num_radial)
    for _ in range(40):
        e2, f2 = energy.fused.energy_force(x)
        assert torch.equal(e2, eng) and torch.equal(f2, force)
