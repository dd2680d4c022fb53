"""cProfile of the first-call path (bind + direct launch) of Schnet.make_model(...)(inputs) over distinct resident batches."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.data.packer import BatchPacker
from gcnn_keras_amd.literature import Schnet
import bench
items = [{"name": "node_number", "ragged": True, "dtype": "float32"}, {"name": "node_coordinates", "ragged": True, "dtype": "float32"},
         {"name": "edge_indices", "ragged": True, "dtype": "int64"}]
n = 96
lists = [bench._graph_list(synth.qm9_like_batch(num_graphs=128, seed=1234 + k)) for k in range(n)]
model = Schnet.make_model(depth=3)
model.set_weights(list(synth.schnet_params(seed=7).values()))
packer = BatchPacker(items, index_item="edge_indices", node_item="node_number", slots=n)
res = [packer.pack(g) for g in lists]
for pb in res:
    pb.wait(torch.cuda.current_stream())
ins = [[pb["node_number"], pb["node_coordinates"], pb["edge_indices"]] for pb in res]
for x in ins[:32]:
    model(x)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for x in ins[32:]:
    model(x)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
