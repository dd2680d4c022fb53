"""Run-to-run bit equality of the PaiNN energy + force pass (config 3) over many calls; which rows move and by how much."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.literature import PAiNN
from gcnn_keras_amd.model.force import EnergyForceModel
from helpers import mol_inputs, painn_weight_list

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = synth.md17_like_batch(num_graphs=64, seed=2345)
p = synth.painn_params(seed=8, random_bias=True)
energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
energy.set_weights(painn_weight_list(p))
model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_as_dict=True, output_to_tensor=False,
                         output_squeeze_states=True)
x = mol_inputs(b)
first = model(x)["force"].values.clone()
torch.cuda.synchronize()
slot = energy.fused.slot_of(x, grad=True)
print("reverse tiles:", None if slot.tiles1 is None else {k: v for k, v in slot.tiles1.items() if k != "table"})
bad = 0
names = ("g_d", "g_rij", "g_s", "gv", "g_zp", "g_vp")
snap = lambda: [getattr(slot, k).clone() for k in names]
model(x); torch.cuda.synchronize()
ref = snap()
for i in range(calls):
    f = model(x)["force"].values
    torch.cuda.synchronize()
    now = snap()
    for k, a_, b_ in zip(names, now, ref):
        if not torch.equal(a_, b_) and bad < 8:
            d = (a_ != b_)
            idx = d.reshape(-1).nonzero().flatten()
            print("   call %d: %s differs in %d elements (first flat idx %s), max |diff| %.3g, scale %.3g" % (
                i, k, int(d.sum()), idx[:5].tolist(), float((a_ - b_).abs().max()), float(b_.abs().max())))
    if not torch.equal(f, first):
        bad += 1
        rows = (f != first).any(1).nonzero().flatten()
        if bad <= 8:
            print("call %d (%s): %d rows differ, first %s, max |diff| %.3g (scale %.3g)" % (
                i, energy.fused.last, len(rows), rows[:6].tolist(), float((f - first).abs().max()), float(first.abs().max())))
print("%d of %d calls differ from the first" % (bad, calls))
