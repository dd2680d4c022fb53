"""Run-to-run bit equality of the fused PaiNN forward / energy + force pass (config 3) over many eager and replayed calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.literature import PAiNN
from gcnn_keras_amd.model.force import EnergyForceModel
from helpers import mol_inputs, painn_weight_list
bad = 0
for seed in (2345, 2346, 2347):
    b = synth.md17_like_batch(num_graphs=64, seed=seed)
    p = synth.painn_params(seed=8, random_bias=True)
    energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
    energy.set_weights(painn_weight_list(p))
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False, output_squeeze_states=True)
    x = mol_inputs(b)
    energy.fused.mode = "eager"
    ref = model(x)
    e0, f0 = ref["energy"].clone(), ref["force"].values.clone()
    for mode, n in (("eager", 30), ("auto", 60)):
        energy.fused.mode = mode
        for i in range(n):
            out = model(x)
            if not (torch.equal(out["energy"], e0) and torch.equal(out["force"].values, f0)):
                bad += 1
                d = (out["force"].values - f0).abs().amax(dim=1)
                print("seed %d mode %s call %d: %d force rows differ, max %.3g; energy rows differ %d" % (
                    seed, mode, i, int((d > 0).sum()), float(d.max()), int((out["energy"] != e0).sum())))
print("mismatching calls:", bad)
