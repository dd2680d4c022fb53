import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.literature import Schnet
from gcnn_keras_amd.model.force import EnergyForceModel
from helpers import dev
from oracle import kgcnn_oracle as ko

b = synth.qm9_like_batch(num_graphs=12, seed=77)
rng = np.random.default_rng(1)
idx, es = b["edge_indices"].copy(), b["edge_splits"]
for g in range(len(es) - 1):
    idx[es[g]:es[g + 1]] = idx[es[g]:es[g + 1]][rng.permutation(es[g + 1] - es[g])]
p = synth.schnet_params(seed=3, random_bias=True)
def E(dt, ii):
    return ko.schnet_forward(ko.to_dtype(p, dt), ko.R(b["node_number"], b["node_splits"]), ko.R(b["node_coordinates"].astype(dt), b["node_splits"]), ko.R(ii, es), depth=3)
for name, ii in (("sorted", b["edge_indices"]), ("unsorted", idx)):
    e64 = E(np.float64, ii).ravel()
    energy = Schnet.make_model(depth=3)
    energy.set_weights(list(p.values()))
    inputs = [dev(b["node_number"], b["node_splits"]), dev(b["node_coordinates"], b["node_splits"]), dev(ii, es)]
    model = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False, output_squeeze_states=True)
    fwd = energy(inputs).cpu().numpy().ravel()
    fwd_layers = energy(inputs, fused=False).cpu().numpy().ravel()
    ef = model(inputs)["energy"].cpu().numpy().ravel()
    ef2 = model(inputs)["energy"].cpu().numpy().ravel()
    model.fused = False
    tape = model(inputs)["energy"].cpu().numpy().ravel()
    o32 = E(np.float32, ii).ravel()
    np.set_printoptions(linewidth=200, precision=2)
    print(name)
    for nm, v in (("oracle32", o32), ("forward route", fwd), ("layer path", fwd_layers), ("force route (direct)", ef), ("force route (graph)", ef2), ("tape", tape)):
        print("  %-22s" % nm, np.abs(v - e64) / np.abs(e64))
