"""BASELINE config 3 (PaiNN, MD17-shaped batch of 64 aspirin-sized graphs): forward and energy+force through the layer
path - eager, replayed from one HIP graph, and with several batches in flight (GraphedModelPool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.engine import GraphedModel, GraphedModelPool
from gcnn_keras_amd.literature import PAiNN
from gcnn_keras_amd.model.force import EnergyForceModel
from gcnn_keras_amd.ragged import RaggedTensor


def inputs_of(b):
    return [RaggedTensor.from_numpy(b["node_number"], b["node_splits"]),
            RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
            RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]


def timeit(fn, n):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batches = [synth.md17_like_batch(num_graphs=graphs, seed=2345 + k) for k in range(4)]
m = int(batches[0]["edge_splits"][-1])
energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
force = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_as_dict=True,
                         output_to_tensor=False, output_squeeze_states=True)
for name, model in (("forward", energy), ("energy+force", force)):
    ins = [inputs_of(b) for b in batches]
    with torch.set_grad_enabled(name != "forward"):
        t_eager = timeit(lambda i: model(ins[0]), 10)
    one = GraphedModel(model, ins[0])
    t_graph = timeit(lambda i: one(), 50)
    line = "PaiNN %s, %d graphs (M=%d): eager %.2f ms, HIP-graph replay %.3f ms" % (name, graphs, m, t_eager * 1e3, t_graph * 1e3)
    for k in (2, 3, 4):
        pool = GraphedModelPool(model, ins[:k])
        t_pool = timeit(pool.replay, 60)
        line += ", %d in flight %.3f ms" % (k, t_pool * 1e3)
        del pool
    print(line + "  (-> %.1f M edges/s at best)" % (m / min(t_graph, t_pool) / 1e6))
