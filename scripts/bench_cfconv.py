"""Micro-benchmark of the fused cfconv kernel alone (HIP events), small (config 2) and large (config 4 shard) edge counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gcnn_keras_amd import _ffi, synth
if os.environ.get("MP_LIB"):  # A/B runs of two builds of the engine on the same box: MP_LIB=path/to/variant.so
    _ffi.LIB_PATH = os.path.abspath(os.environ["MP_LIB"])
from gcnn_keras_amd.engine import _HipTimer
from gcnn_keras_amd.fused import FusedSchnet

def run(graphs, flags_list=(1, 5, 9), iters=100):
    b = synth.qm9_like_batch(num_graphs=graphs, seed=1234)
    p = synth.schnet_params(seed=7)
    for fl in flags_list:
        f = FusedSchnet(p, depth=3, fast_softplus=bool(fl & 1), cfconv_flags=fl & 6)
        dev = {"z": torch.from_numpy(b["node_number"]).cuda(), "xyz": torch.from_numpy(b["node_coordinates"]).cuda(),
               "idx": torch.from_numpy(b["edge_indices"]).cuda(), "ns": torch.from_numpy(b["node_splits"]).cuda(),
               "es": torch.from_numpy(b["edge_splits"]).cuda(), "ns_host": b["node_splits"], "es_host": b["edge_splits"]}
        n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
        f.bind(dev, n, m, graphs)
        f.forward(); torch.cuda.synchronize()
        r = f.roofline(8000.0, 157.3, iters)
        with torch.cuda.stream(f.stream):
            ms_fwd = _HipTimer().time_ms(lambda: _ffi.call("mp_graph_launch", f.graph, _ffi.stream()), iters)
        print("graphs=%d N=%d M=%d flags=%d cfconv %.1f us  %.1f TF (%.1f%%)  forward %.1f us  %.1f Medges/s" % (
            graphs, n, m, fl, r["avg_launch_us"], r["achieved"], 100 * r["frac"], ms_fwd * 1e3, m / ms_fwd / 1e3))

if __name__ == "__main__":
    sizes = [int(s) for s in sys.argv[1:]] or [128, 12500]
    for g in sizes:
        run(g)


def diag(graphs):
    b = synth.qm9_like_batch(num_graphs=graphs, seed=1234)
    p = synth.schnet_params(seed=7)
    f = FusedSchnet(p, depth=3, fast_softplus=True)
    dev = {"z": torch.from_numpy(b["node_number"]).cuda(), "xyz": torch.from_numpy(b["node_coordinates"]).cuda(),
           "idx": torch.from_numpy(b["edge_indices"]).cuda(), "ns": torch.from_numpy(b["node_splits"]).cuda(),
           "es": torch.from_numpy(b["edge_splits"]).cuda(), "ns_host": b["node_splits"], "es_host": b["edge_splits"]}
    n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
    f.bind(dev, n, m, graphs)
    f.forward(); torch.cuda.synchronize()
    d = torch.zeros(8, dtype=torch.int64, device="cuda")
    scratch = torch.zeros_like(f.agg)
    def launch():
        _ffi.call("mp_cfconv_gauss_diag_f32", _ffi.ptr(f.x), n, _ffi.ptr(f.dist), 20, 4.0, 0.4, 0.0,
                  _ffi.ptr(f.packed[0]), _ffi.ptr(f.recv), _ffi.ptr(f.send), None, m, _ffi.ptr(scratch),
                  _ffi.ptr(d), _ffi.stream())
    ms = _HipTimer().time_ms(launch, 10)
    d.zero_()
    launch()
    torch.cuda.synchronize()
    waves = min(256, ((m + 31) // 32 + 3) // 4) * 4
    cyc = d.cpu().numpy().astype(np.float64).sum() / waves
    print("diag kernel %.1f us, %.0f shader cycles per wave -> clock %.2f GHz" % (ms * 1e3, cyc, cyc / (ms * 1e3) / 1e3))
    v = d.cpu().numpy().astype(np.float64)
    names = ["stage", "setup+gauss", "gemm1", "ssp+xload", "gemm2", "seg-accumulate", "tail-xchg", "boundary-flush"]
    ntiles = (m + 31) // 32
    print("diag graphs=%d tiles=%d: " % (graphs, ntiles) + "  ".join("%s %.1f%%" % (nm, 100 * x / v.sum()) for nm, x in zip(names, v)))
    print("   cycles per tile: " + "  ".join("%s %.0f" % (nm, x / ntiles) for nm, x in zip(names[1:], v[1:])),
          " stage per WG %.0f" % (v[0] / max(1, min(256, (ntiles + 3) // 4)) / 4))


if __name__ == "__main__" and os.environ.get("MP_DIAG"):
    for g in sizes:
        diag(g)
