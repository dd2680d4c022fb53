"""Throughput of the fused SchNet forward with several forwards in flight (one HIP stream + graph + buffer set each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcnn_keras_amd import _ffi, synth
if os.environ.get("MP_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["MP_LIB"])
from gcnn_keras_amd.fused import FusedSchnet


def make(graphs, k, cf=0):
    b = synth.qm9_like_batch(num_graphs=graphs, seed=1234)
    p = synth.schnet_params(seed=7)
    out = []
    for _ in range(k):
        f = FusedSchnet(p, depth=3, fast_softplus=True, cfconv_flags=cf)
        dev = {"z": torch.from_numpy(b["node_number"]).cuda(), "xyz": torch.from_numpy(b["node_coordinates"]).cuda(),
               "idx": torch.from_numpy(b["edge_indices"]).cuda(), "ns": torch.from_numpy(b["node_splits"]).cuda(),
               "es": torch.from_numpy(b["edge_splits"]).cuda(), "ns_host": b["node_splits"], "es_host": b["edge_splits"]}
        n, m = int(b["node_splits"][-1]), int(b["edge_splits"][-1])
        f.bind(dev, n, m, graphs)
        f.forward(); torch.cuda.synchronize()
        out.append(f)
    return out, m


def run(graphs, k, steps=400, cf=0):
    fs, m = make(graphs, k, cf)
    ref = fs[0].forward().clone(); torch.cuda.synchronize()
    def loop(n):
        for i in range(n):
            fs[i % k].replay()
    loop(40); torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / steps * 1e6)
    dt = min(res) * steps / 1e6
    print("   reps us/step:", " ".join("%.1f" % r for r in res))
    same = all(torch.equal(f.out, ref) for f in fs)
    print("cfconv_flags=%d graphs=%d in_flight=%d: %.1f us/step  %.1f Medges/s  outputs identical: %s" % (cf, graphs, k, dt / steps * 1e6, m * steps / dt / 1e6, same))


if __name__ == "__main__":
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    for cf in (16, 4, 0):
        for k in (1, 3, 4):
            run(g, k, cf=cf)
