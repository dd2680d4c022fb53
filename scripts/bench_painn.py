"""BASELINE config 3 (PaiNN, MD17-shaped batch of 64 aspirin-sized graphs, energy + forces) on one MI355X.

    python scripts/bench_painn.py [graphs] [--no-layers]

Prints ONE JSON line: latency of ``PAiNN.make_model(...)(inputs)`` (fused pipeline, HIP-graph replay) and of
``EnergyForceModel(...)(inputs)`` (fused forward + hand-written reverse pass, one graph), the same through the layer path
replayed from a graph (round-1 route, for comparison), throughput with several batches in flight, and per-kernel times of
the three kernel classes (message, message reverse, GEMM) measured with HIP events on the stream they run on, with the
roofline that bounds each.  A parity configuration, not the headline bench line (that is bench.py, config 2)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.engine import GraphedModel, _HipTimer
from gcnn_keras_amd.literature import PAiNN
from gcnn_keras_amd.model.force import EnergyForceModel
from gcnn_keras_amd.ragged import RaggedTensor

HBM_PEAK, MFMA_PEAK = 8000.0, 157.3   # GB/s, TFLOP/s (MI355X_MICROARCH.md)


def inputs_of(b):
    return [RaggedTensor.from_numpy(b["node_number"], b["node_splits"]),
            RaggedTensor.from_numpy(b["node_coordinates"], b["node_splits"]),
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


def painn_flops(n, m, f=128, b=20, d=3):
    """Forward: per block five GEMMs (30 N F^2) + the per-edge filter and products (M (6 B F + 12 F))."""
    return d * (30 * n * f * f + m * (6 * b * f + 12 * f))


def main():
    graphs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
    with_layers = "--no-layers" not in sys.argv
    batches = [synth.md17_like_batch(num_graphs=graphs, seed=2345 + k) for k in range(4)]
    n, m = int(batches[0]["node_splits"][-1]), int(batches[0]["edge_splits"][-1])
    energy = PAiNN.make_model(equiv_initialize_kwargs={"dim": 3, "method": "eps"})
    force = EnergyForceModel(model_energy=energy, coordinate_input=1, energy_output=0, output_as_dict=True,
                             output_to_tensor=False, output_squeeze_states=True)
    ins = [inputs_of(b) for b in batches]
    out = {"workload": "BASELINE config 3: PAiNN.make_model (F=128, depth 3, Bessel 20) on %d MD17-shaped graphs, N=%d, "
                       "M=%d; energy (G,1) and forces (N,3)" % (graphs, n, m), "graphs": graphs, "nodes": n, "edges": m}
    for k in range(4):                       # bind + capture every batch (outside the timings)
        energy(ins[k]), energy(ins[k]), force(ins[k]), force(ins[k])
    torch.cuda.synchronize()
    t_fwd = timeit(lambda i: energy(ins[0]), 200)
    t_ef = timeit(lambda i: force(ins[0]), 200)
    out["fused"] = {"forward_ms": t_fwd * 1e3, "energy_force_ms": t_ef * 1e3,
                    "forward_edges_per_s": m / t_fwd, "energy_force_edges_per_s": m / t_ef,
                    "forward_mfma_frac": painn_flops(n, m) / t_fwd / (MFMA_PEAK * 1e12)}
    streams = [torch.cuda.Stream() for _ in range(4)]
    for k in (2, 4):
        def step(i, k=k):
            with torch.cuda.stream(streams[i % k]):
                force(ins[i % k])
        t = timeit(step, 200)
        out["fused"]["energy_force_ms_%d_in_flight" % k] = t * 1e3
        def stepf(i, k=k):
            with torch.cuda.stream(streams[i % k]):
                energy(ins[i % k])
        out["fused"]["forward_ms_%d_in_flight" % k] = timeit(stepf, 200) * 1e3
    # launch groups: the four batches concatenated on the device and served by one launch sequence (route.call_group)
    for _ in range(3):
        energy.fused.call_group(ins, with_forces=True), energy.fused.call_group(ins)
    torch.cuda.synchronize()
    t_gf = timeit(lambda i: energy.fused.call_group(ins), 100) / 4
    t_gef = timeit(lambda i: energy.fused.call_group(ins, with_forces=True), 100) / 4
    out["fused"]["forward_ms_per_batch_in_a_group_of_4"] = t_gf * 1e3
    out["fused"]["energy_force_ms_per_batch_in_a_group_of_4"] = t_gef * 1e3
    if with_layers:
        force.fused = False
        g_f = GraphedModel(lambda x: energy(x, fused=False), ins[0], grad=False)
        g_ef = GraphedModel(force, ins[0])
        out["layer_path_graph_replay"] = {"forward_ms": timeit(lambda i: g_f(), 50) * 1e3,
                                          "energy_force_ms": timeit(lambda i: g_ef(), 50) * 1e3}
        force.fused = None
    # kernel classes, timed alone with HIP events on the stream they are launched on
    force(ins[0])                            # (the group's union may have taken this batch's place in the slot table)
    slot = energy.fused.slot_of(ins[0], grad=True)
    p, w, blk = slot.p, slot.w, slot.blk[1]
    timer = _HipTimer()
    v_in = slot.vs[0]

    def msg():
        _ffi.call("mp_painn_message_f32", _ffi.ptr(blk["s"]), _ffi.ptr(v_in), n, _ffi.ptr(slot.rbf), slot.B, None,
                  _ffi.ptr(slot.rij), _ffi.ptr(p["conv1/w/kernel"]), _ffi.ptr(p["conv1/w/bias"]), _ffi.ptr(slot.ptr0),
                  _ffi.ptr(slot.perm0), _ffi.ptr(slot.send), m, _ffi.ptr(slot.zs[0]), _ffi.ptr(blk["zp"]),
                  _ffi.ptr(blk["vp"]), _ffi.stream())

    def msg_bwd():
        _ffi.call("mp_painn_message_bwd_f32", _ffi.ptr(blk["s"]), _ffi.ptr(v_in), n, _ffi.ptr(slot.rbf),
                  _ffi.ptr(slot.rbfd), slot.B, None, None, _ffi.ptr(slot.rij), _ffi.ptr(p["conv1/w/kernel"]),
                  _ffi.ptr(p["conv1/w/bias"]), _ffi.ptr(slot.ptr1), _ffi.ptr(slot.perm1), _ffi.ptr(slot.recv), m,
                  _ffi.ptr(slot.g_zp), _ffi.ptr(slot.g_vp), _ffi.ptr(slot.g_s), _ffi.ptr(slot.gv), _ffi.ptr(slot.g_d),
                  _ffi.ptr(slot.g_rij), 0, _ffi.stream())

    def msg_tiles():
        tl = slot.tiles0
        _ffi.call("mp_painn_message_tiles_f32", _ffi.ptr(blk["s"]), _ffi.ptr(v_in), n, _ffi.ptr(slot.rbf), slot.B, None,
                  _ffi.ptr(slot.rij), _ffi.ptr(w["conv1/w/F"]), _ffi.ptr(slot.ptr0), _ffi.ptr(slot.send), m,
                  _ffi.ptr(tl["table"]), tl["count"], tl["max_rows"], tl["max_edges"], _ffi.ptr(slot.zs[0]),
                  _ffi.ptr(blk["zp"]), _ffi.ptr(blk["vp"]), _ffi.stream())

    def msg_bwd_tiles():
        tl = slot.tiles1
        _ffi.call("mp_painn_message_bwd_tiles_f32", _ffi.ptr(blk["s"]), _ffi.ptr(v_in), n, _ffi.ptr(slot.rbf),
                  _ffi.ptr(slot.rbfd), slot.B, None, None, _ffi.ptr(slot.rij), _ffi.ptr(w["conv1/w/F"]), _ffi.ptr(slot.ptr1),
                  _ffi.ptr(slot.perm1), _ffi.ptr(slot.recv), m, _ffi.ptr(tl["table"]), tl["count"], tl["max_rows"],
                  tl["max_own"], tl["max_edges"], _ffi.ptr(slot.g_zp), _ffi.ptr(slot.g_vp), _ffi.ptr(slot.g_s),
                  _ffi.ptr(slot.gv), _ffi.ptr(slot.g_d), _ffi.ptr(slot.g_rij), 0, _ffi.stream())

    def gemm():
        slot._chain(blk["vp"], 3 * n, 128, w["uv1/P"], None, 256, blk["uv"])

    def gemm2():
        slot._chain(slot.zs[0], n, 128, w["conv1/dense1/P"], p.get("conv1/dense1/bias"), 128, blk["s"], act=slot.act_conv,
                    save_pre=blk["h1"], w2=w["conv1/phi/P"], b2=p.get("conv1/phi/bias"), u2=384)

    f = 128
    kernels = {}
    tiled = []
    if slot.tiles0 is not None:      # the default kernels of a receiver-sorted batch: tiles in LDS, filter on the matrix pipe
        tiled.append(("painn_message_tile_kernel (default)", msg_tiles, m * (6 * 20 * f + 12 * f),
                      4 * (3 * n * f * 2 + 4 * n * f * 2) + m * (4 * 20 + 24), "hbm"))
    if slot.tiles1 is not None:
        tiled.append(("painn_message_bwd_tile_kernel (default)", msg_bwd_tiles, m * (12 * 20 * f + 30 * f),
                      4 * (3 * n * f * 2 + 4 * n * f + 6 * n * f) + m * (8 * 20 + 40), "hbm"))
    for name, fn, flops, nbytes, bound in tiled + [
            ("painn_message_kernel (gather route)", msg, m * (6 * 20 * f + 12 * f), 4 * (3 * n * f * 2 + 4 * n * f * 2) + m * (4 * 20 + 24),
             "hbm"),
            ("painn_message_bwd_kernel (sender-parallel route)", msg_bwd, m * (12 * 20 * f + 30 * f),
             4 * (3 * n * f * 2 + 4 * n * f + 6 * n * f) + m * (8 * 20 + 40), "hbm"),
            ("dense_chain_kernel (3N,128)x(128,256)", gemm, 2 * 3 * n * 128 * 256, 4 * (3 * n * 128 + 128 * 256 + 3 * n * 256),
             "mfma"),
            ("dense_chain_kernel (N,128)x(128,128)x(128,384)", gemm2, 2 * n * 128 * (128 + 384),
             4 * (n * 128 * 2 + 128 * 512 + n * 384), "mfma")]:
        ms = timer.time_ms(fn, 50)
        kernels[name] = {"avg_launch_us": ms * 1e3, "algorithmic_flops": flops, "algorithmic_bytes": nbytes,
                         "bound": bound, "tflops": flops / (ms * 1e-3) / 1e12, "gbs": nbytes / (ms * 1e-3) / 1e9,
                         "frac": (flops / (ms * 1e-3) / 1e12 / MFMA_PEAK) if bound == "mfma"
                         else (nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK)}
    out["kernels"] = kernels
    print(json.dumps(out))


if __name__ == "__main__":
    main()
