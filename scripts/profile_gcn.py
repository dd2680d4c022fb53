"""The GCN forward of BASELINE config 5 replayed from the model's own HIP graph - the workload behind
profiles/r02_gcn_config5_kernel_stats.csv:
    rocprofv3 --kernel-trace --stats -d <dir> -o gcn -- python3 scripts/profile_gcn.py [replays]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcnn_keras_amd import synth
from gcnn_keras_amd.literature import GCN
from gcnn_keras_amd.ragged import RaggedTensor

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = synth.cora_like_graph()
f = int(g["node_attributes"].shape[1])
ins = [RaggedTensor.from_numpy(g["node_attributes"], g["node_splits"]),
       RaggedTensor.from_numpy(g["edge_weights"], g["edge_splits"]),
       RaggedTensor.from_numpy(g["edge_indices"], g["edge_splits"])]
model = GCN.make_model(inputs=[{"shape": (None, f), "name": "node_attributes", "dtype": "float32", "ragged": True},
                               {"shape": (None, 1), "name": "edge_weights", "dtype": "float32", "ragged": True},
                               {"shape": (None, 2), "name": "edge_indices", "dtype": "int64", "ragged": True}],
                       gcn_args={"units": 64, "use_bias": True, "activation": "relu", "pooling_method": "sum"},
                       depth=3, output_embedding="node",
                       output_mlp={"use_bias": [True, True, False], "units": [64, 32, 7],
                                   "activation": ["relu", "relu", "softmax"]})
for _ in range(replays):
    model(ins)
torch.cuda.synchronize()
