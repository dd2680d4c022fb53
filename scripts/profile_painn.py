"""One PaiNN energy+force batch (BASELINE config 3) replayed alone - the workload behind profiles/r02_painn_*:
    rocprofv3 --kernel-trace --stats -d <dir> -o painn -- python3 scripts/profile_painn.py [forward|force] [replays]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcnn_keras_amd import synth
from gcnn_keras_amd.literature import PAiNN
from gcnn_keras_amd.model.force import EnergyForceModel
from gcnn_keras_amd.ragged import RaggedTensor

what = sys.argv[1] if len(sys.argv) > 1 else "force"
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 200
b = synth.md17_like_batch(num_graphs=64, seed=2345)
ins = [RaggedTensor.from_numpy(b["node_number"], b["node_splits"]),
       RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
       RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]
energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
model = energy if what == "forward" else EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0,
                                                          output_to_tensor=False, output_squeeze_states=True)
for _ in range(replays):
    model(ins)
torch.cuda.synchronize()
