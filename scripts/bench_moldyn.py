"""MD inference step (kgcnn/moldyn/base.py:106-165, one molecule, energy + forces): latency of
``MolDynamicsModelPredictor.__call__`` on one 21-atom MD17-shaped molecule with a PaiNN ``EnergyForceModel``, eager and
with ``use_graph=True``; ``--profile`` adds a cProfile table of the host side of the replayed step.

    python scripts/bench_moldyn.py [--profile] [steps]
"""
import cProfile
import json
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from gcnn_keras_amd import synth
from gcnn_keras_amd.moldyn import MolDynamicsModelPredictor
from test_gpu_moldyn import ITEMS, _graphs, _painn_ef


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = int(args[0]) if args else 300
    model = _painn_ef()
    b = synth.md17_like_batch(num_graphs=1, seed=6)
    outs = {"energy": "energy", "forces": "force"}
    eager = MolDynamicsModelPredictor(model=model, model_inputs=ITEMS, model_outputs=outs)
    fast = MolDynamicsModelPredictor(model=model, model_inputs=ITEMS, model_outputs=outs, use_graph=True)
    rng = np.random.default_rng(0)
    xyz = b["node_coordinates"].copy()

    def run(pred, n):
        x = xyz
        for _ in range(n):
            x = x + rng.normal(scale=0.001, size=x.shape).astype(np.float32)
            pred(_graphs(b, x))

    run(eager, 5), run(fast, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(eager, 30); torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / 30
    t0 = time.perf_counter(); run(fast, steps); torch.cuda.synchronize(); t_fast = (time.perf_counter() - t0) / steps
    print(json.dumps({"workload": "MD step: PaiNN energy + forces, one 21-atom molecule, N=%d, M=%d"
                                  % (int(b["node_splits"][-1]), int(b["edge_splits"][-1])),
                      "step_ms_eager": t_eager * 1e3, "step_ms_graph_replay": t_fast * 1e3,
                      "graph_captures": fast.graph_captures}))
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable(); run(fast, steps); pr.disable()
        st = pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative")
        st.print_stats(45)


if __name__ == "__main__":
    main()
