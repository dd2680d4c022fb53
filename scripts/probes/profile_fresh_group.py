"""Host-side profile of the never-seen-batch legs of bench.py --workload stream (cProfile around the timed loops)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from gcnn_keras_amd import synth
from gcnn_keras_amd.data.packer import BatchPacker
from gcnn_keras_amd.literature import Schnet


def main():
    batches, in_flight, grp = 64, 4, 5
    items = [{"name": "node_number", "ragged": True, "dtype": "float32"},
             {"name": "node_coordinates", "ragged": True, "dtype": "float32"},
             {"name": "edge_indices", "ragged": True, "dtype": "int64"}]
    lists, edges = [], []
    for k in range(batches):
        b = synth.qm9_like_batch(num_graphs=128, seed=1234 + k)
        lists.append(bench._graph_list(b))
        edges.append(int(b["edge_splits"][-1]))
    model = Schnet.make_model(depth=3)
    model.set_weights(list(synth.schnet_params(seed=7).values()))
    streams = [torch.cuda.Stream() for _ in range(in_flight)]
    packer = BatchPacker(items, index_item="edge_indices", node_item="node_number", slots=batches)
    resident = [packer.pack(g) for g in lists]
    for pb in resident:
        pb.wait(torch.cuda.current_stream())
    ins = [[pb["node_number"], pb["node_coordinates"], pb["edge_indices"]] for pb in resident]
    base = torch.cuda.current_stream()

    def grouped():
        res = []
        for k in range(0, batches - batches % grp, grp):
            torch.cuda.set_stream(streams[(k // grp) % in_flight])
            res.extend(model.fused.call_group(ins[k:k + grp]))
        torch.cuda.set_stream(base)
        return res

    def single():
        res = []
        for k, x in enumerate(ins):
            torch.cuda.set_stream(streams[k % in_flight])
            res.append(model(x))
        torch.cuda.set_stream(base)
        return res

    for name, fn in (("grouped", grouped), ("single", single)):
        for w in range(2):
            fn()
            torch.cuda.synchronize()
            model.fused.release() if w == 0 else None
        model.fused._groups.clear()
        model.fused._slots.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("%s: host %.3f ms, with drain %.3f ms -> %.0f M edges/s" % (name, t_host * 1e3, t_all * 1e3,
                                                                         sum(edges) / t_all / 1e6))
        del r
        model.fused._groups.clear()
        model.fused._slots.clear()
        torch.cuda.synchronize()
        print("arena before profiled pass: made %d taken %d" % (model.fused._arena.made, model.fused._arena.taken))
        pr = cProfile.Profile()
        pr.enable()
        r = fn()
        pr.disable()
        torch.cuda.synchronize()
        del r
        print("arena after profiled pass: made %d taken %d" % (model.fused._arena.made, model.fused._arena.taken))
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
        model.fused.release()


if __name__ == "__main__":
    main()
