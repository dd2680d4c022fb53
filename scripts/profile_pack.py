"""cProfile of BatchPacker.pack + PackedBatch.wait for 128-graph batches."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.data.packer import BatchPacker
import bench
items = [{"name": "node_number", "ragged": True, "dtype": "float32"}, {"name": "node_coordinates", "ragged": True, "dtype": "float32"},
         {"name": "edge_indices", "ragged": True, "dtype": "int64"}]
lists = [bench._graph_list(synth.qm9_like_batch(num_graphs=128, seed=1234 + k)) for k in range(40)]
packer = BatchPacker(items, index_item="edge_indices", node_item="node_number", slots=8)
s = torch.cuda.Stream()
for g in lists[:8]:
    packer.pack(g).wait(s)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for g in lists[8:]:
    packer.pack(g).wait(s)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
