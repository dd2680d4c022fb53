"""SchNet energy + forces on one MI355X: the fork's force_schnet.py model (int64 numbers, embedding 128, depth 6,
Gauss(25, 5.0, 0.4), last_mlp [128, 64, 1], no output MLP) on 64 MD17-shaped graphs, and the reference default head
(depth 3, QM9-shaped graphs).

    python scripts/bench_schnet_force.py [graphs] [--profile fork|default [replays]]

Prints ONE JSON line: latency of ``EnergyForceModel(...)(inputs)`` through the fused route (forward + hand-written
reverse pass, one HIP graph), with 4 batches in flight, the tape + layer path replayed from a graph, and the kernel
classes of the reverse pass timed alone with HIP events on the stream they run on.  With ``--profile`` only replays the
fused pass (the workload behind profiles/r02_schnet_force_*).  A parity configuration, not the headline bench line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.engine import GraphedModel, _HipTimer
from gcnn_keras_amd.literature import Schnet
from gcnn_keras_amd.model.force import EnergyForceModel
from gcnn_keras_amd.ragged import RaggedTensor

HBM_PEAK, MFMA_PEAK = 8000.0, 157.3   # GB/s, TFLOP/s (MI355X_MICROARCH.md)

FORK = dict(
    inputs=[{"shape": [None], "name": "node_number", "dtype": "int64", "ragged": True},
            {"shape": [None, 3], "name": "node_coordinates", "dtype": "float32", "ragged": True},
            {"shape": [None, 2], "name": "range_indices", "dtype": "int64", "ragged": True}],
    input_embedding={"node": {"input_dim": 95, "output_dim": 128}},
    interaction_args={"units": 128, "use_bias": True, "activation": "shifted_softplus", "cfconv_pool": "sum"},
    node_pooling_args={"pooling_method": "sum"}, depth=6,
    gauss_args={"bins": 25, "distance": 5, "offset": 0.0, "sigma": 0.4}, verbose=10,
    last_mlp={"use_bias": [True] * 3, "units": [128, 64, 1], "activation": ["shifted_softplus"] * 2 + ["linear"]},
    output_embedding="graph", output_to_tensor=True, use_output_mlp=False, output_mlp=None)


def inputs_of(b, int64):
    z = b["node_number"].astype(np.int64) if int64 else b["node_number"]
    return [RaggedTensor.from_numpy(z, b["node_splits"]), RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
            RaggedTensor.from_numpy(b["edge_indices"], b["edge_splits"])]


def timeit(fn, n, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def case(which, graphs):
    if which == "fork":
        batches = [synth.md17_like_batch(num_graphs=graphs, seed=2345 + k) for k in range(4)]
        energy = Schnet.make_model(**FORK)
        force = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False,
                                 output_squeeze_states=True, is_physical_force=False, output_as_dict=False)
        return batches, energy, force, [inputs_of(b, True) for b in batches], 6, 25
    batches = [synth.qm9_like_batch(num_graphs=graphs, seed=2345 + k) for k in range(4)]
    energy = Schnet.make_model(depth=3)
    force = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_to_tensor=False,
                             output_squeeze_states=True)
    return batches, energy, force, [inputs_of(b, False) for b in batches], 3, 20


def measure(which, graphs, with_layers):
    batches, energy, force, ins, depth, bins = case(which, graphs)
    n, m = int(batches[0]["node_splits"][-1]), int(batches[0]["edge_splits"][-1])
    out = {"graphs": graphs, "nodes": n, "edges": m, "depth": depth}
    for k in range(4):
        energy(ins[k]), energy(ins[k]), force(ins[k]), force(ins[k])
    torch.cuda.synchronize()
    t_f = timeit(lambda i: energy(ins[0]), 200)
    t_ef = timeit(lambda i: force(ins[0]), 200)
    out["fused"] = {"forward_ms": t_f * 1e3, "energy_force_ms": t_ef * 1e3, "energy_force_edges_per_s": m / t_ef}
    streams = [torch.cuda.Stream() for _ in range(4)]

    def step(i):
        with torch.cuda.stream(streams[i % 4]):
            force(ins[i % 4])
    out["fused"]["energy_force_ms_4_in_flight"] = timeit(step, 200) * 1e3
    if with_layers:
        force.fused = False
        g_ef = GraphedModel(force, ins[0])
        out["layer_path_graph_replay"] = {"energy_force_ms": timeit(lambda i: g_ef(), 50) * 1e3}
        force.fused = None
    slot = next(iter(energy.fused._gslots.values()))
    timer = _HipTimer()
    ga = slot.gauss
    blk = depth - 1
    T, cf_img = slot.gw["T"], slot.w["cfconv"][blk]
    seg0 = slot.recv if slot.perm0 is None else slot.seg0
    seg1 = slot.send if slot.perm1 is None else slot.seg1
    gargs = (int(ga["bins"]), float(ga["distance"]), float(ga["sigma"]), float(ga["offset"]))

    def cf():
        _ffi.call("mp_cfconv_gauss_fused_f32", _ffi.ptr(slot.xs[blk]), n, _ffi.ptr(slot.dist), *gargs, _ffi.ptr(cf_img),
                  _ffi.ptr(seg0), _ffi.ptr(slot.send), _ffi.ptr(slot.perm0), m, slot.flags_arg, _ffi.ptr(scratch),
                  _ffi.stream())

    def cf_swapped():
        _ffi.call("mp_cfconv_gauss_fused_f32", _ffi.ptr(slot.g_agg), n, _ffi.ptr(slot.dist), *gargs, _ffi.ptr(cf_img),
                  _ffi.ptr(seg1), _ffi.ptr(slot.recv), _ffi.ptr(slot.perm1), m, slot.flags_arg, _ffi.ptr(scratch),
                  _ffi.stream())

    def dgrad():
        _ffi.call("mp_cfconv_gauss_dist_grad_f32", _ffi.ptr(slot.xs[blk]), _ffi.ptr(slot.g_agg), n, _ffi.ptr(slot.dist),
                  *gargs, _ffi.ptr(slot.gw["cf"][blk]), _ffi.ptr(slot.recv), _ffi.ptr(slot.send), m, 0,
                  _ffi.ptr(scratch_d), _ffi.stream())

    pre = "interaction%d/" % blk

    def chain():
        _ffi.call("mp_schnet_bwd_block_f32", _ffi.ptr(scratch), n, _ffi.ptr(T[pre + "dense1/kernel"]), _ffi.ptr(scratch_n),
                  _ffi.ptr(T["interaction%d/dense3/kernel" % (blk - 1)]), _ffi.ptr(slot.d2[blk - 1]),
                  _ffi.ptr(T["interaction%d/dense2/kernel" % (blk - 1)]), _ffi.ptr(scratch_a), _ffi.stream())

    scratch, scratch_n, scratch_a = (torch.zeros(n, 128, device="cuda") for _ in range(3))
    scratch_d = torch.zeros(m, device="cuda")
    f = 128
    fl_fwd = m * (2 * bins * f + 2 * f * f + 2 * f)            # two filter GEMMs + the product
    fl_dg = m * (2 * 2 * bins * f + 2 * 2 * f * f + 4 * f)     # basis and derivative chains, GH GEMM, contraction
    by = 4 * (2 * n * f + m * 3)
    kernels = {}
    for name, fn, flops, nbytes in (("cfconv_fused_kernel (receiver side)", cf, fl_fwd, by),
                                    ("cfconv_fused_kernel (columns swapped)", cf_swapped, fl_fwd, by),
                                    ("cfconv_dist_grad_kernel", dgrad, fl_dg, by),
                                    ("schnet_bwd_chain_kernel (block: 3 GEMMs)", chain, 3 * 2 * n * f * f, 4 * 5 * n * f)):
        ms = timer.time_ms(fn, 50)
        kernels[name] = {"avg_launch_us": ms * 1e3, "algorithmic_flops": flops, "algorithmic_bytes": nbytes,
                         "bound": "mfma", "tflops": flops / (ms * 1e-3) / 1e12,
                         "frac": flops / (ms * 1e-3) / 1e12 / MFMA_PEAK}
    out["kernels"] = kernels
    return out


def main():
    graphs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
    if "--profile" in sys.argv:
        k = sys.argv.index("--profile")
        which = sys.argv[k + 1]
        replays = int(sys.argv[k + 2]) if len(sys.argv) > k + 2 else 200
        _, _, force, ins, _, _ = case(which, graphs)
        for _ in range(replays):
            force(ins[0])
        torch.cuda.synchronize()
        return
    out = {"workload": "SchNet energy (G,1) + forces (N,3) via EnergyForceModel(Schnet.make_model(...)) on %d graphs"
                       % graphs,
           "fork_force_schnet": measure("fork", graphs, "--no-layers" not in sys.argv),
           "reference_default": measure("default", graphs, "--no-layers" not in sys.argv)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
