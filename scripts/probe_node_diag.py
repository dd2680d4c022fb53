"""Phase breakdown of the SchNet node-update kernel.  Needs the diagnostic library:

    make -C gcnn_keras_amd/csrc diag && python scripts/probe_node_diag.py [graphs]
"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gcnn_keras_amd import _ffi, synth
_ffi.LIB_PATH = os.path.abspath(os.environ.get("MP_LIB", os.path.join(os.path.dirname(_ffi.LIB_PATH), "libmpengine_diag.so")))
from gcnn_keras_amd.engine import _HipTimer

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
b = synth.qm9_like_batch(num_graphs=graphs, seed=1234)
n = int(b["node_splits"][-1])
p = synth.schnet_params(seed=7)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
agg = torch.randn(n, 128, device="cuda"); nn_ = torch.randn(n, 128, device="cuda"); x = torch.empty(n, 128, device="cuda")
W2, b2 = dev(p["interaction0/dense2/kernel"]), dev(p["interaction0/dense2/bias"])
W3, b3 = dev(p["interaction0/dense3/kernel"]), dev(p["interaction0/dense3/bias"])
Wx = dev(p["interaction1/dense1/kernel"])
flags = 1
if len(sys.argv) > 2 and sys.argv[2] == "bf":     # the forward's default build: bf16-piece weight images (flags 2 | 64)
    def pack(w):
        out = torch.empty(w.numel() * 3 // 2, device="cuda")
        _ffi.call("mp_schnet_node_pack_bf16_f32", _ffi.ptr(w), int(w.shape[0]), int(w.shape[1]), _ffi.ptr(out), _ffi.stream())
        return out
    W2, W3, Wx = pack(W2), pack(W3), pack(Wx)
    flags = 1 | 2 | 64
def launch():
    _ffi.call("mp_schnet_node_update_f32", _ffi.ptr(agg), n, _ffi.ptr(W2), _ffi.ptr(b2), _ffi.ptr(W3), _ffi.ptr(b3),
              _ffi.ptr(nn_), _ffi.ptr(Wx), _ffi.ptr(x), flags, _ffi.stream())
ms = _HipTimer().time_ms(launch, 10)
lib = _ffi.lib(); out = (ctypes.c_ulonglong * 8)()
lib.mp_debug_node_diag(out); launch(); torch.cuda.synchronize(); lib.mp_debug_node_diag(out)
v = np.array(list(out), dtype=np.float64)
names = ["stage+barrier", "gemm1", "epi1+barrier", "gemm2", "epi2+barrier", "gemm3", "epi3+barrier", "loop"]
tiles = (n + 15) // 16
print("N=%d node MID %.1f us; cycles per tile (thread 0 of each workgroup):" % (n, ms * 1e3))
print("  " + "  ".join("%s %.0f" % (nm, x_ / tiles) for nm, x_ in zip(names, v)), " total %.0f" % (v.sum() / tiles))
