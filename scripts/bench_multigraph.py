"""Experiment: the k batch slots as k independent branches of ONE HIP graph (one launch per k forwards) instead of k
graphs on k streams."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.engine import SchnetForward

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b = synth.qm9_like_batch(num_graphs=128, seed=1234)
p = synth.schnet_params(seed=7)
fwd = SchnetForward(p, depth=3, mode="fused", in_flight=k)
fwd.load_batch(b)
m = int(b["edge_splits"][-1])
ref = fwd._slots[0].out.clone()

main = torch.cuda.Stream()
sides = [torch.cuda.Stream() for _ in range(k)]
torch.cuda.synchronize()
with torch.cuda.stream(main):
    _ffi.call("mp_graph_begin", _ffi.stream())
    fork = torch.cuda.Event(); fork.record(main)
    joins = []
    for slot, st in zip(fwd._slots, sides):
        st.wait_event(fork)
        with torch.cuda.stream(st):
            slot._launch_all()
            ev = torch.cuda.Event(); ev.record(st); joins.append(ev)
    for ev in joins:
        main.wait_event(ev)
    exe = ctypes.c_void_p()
    _ffi.call("mp_graph_end", _ffi.stream(), ctypes.byref(exe))
torch.cuda.synchronize()
mp = ctypes.c_void_p(main.cuda_stream)
launch = _ffi.lib().mp_graph_launch
for _ in range(20):
    launch(exe, mp)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    launch(exe, mp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (n * k)
ok = all(torch.equal(s.out, ref) for s in fwd._slots)
print("one graph with %d branches: %.1f us per forward (%.1f Medges/s), outputs identical: %s" % (k, dt * 1e6, m / dt / 1e6, ok))
for i in range(40):
    fwd.replay(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n * k):
    fwd.replay(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (n * k)
print("%d graphs on %d streams    : %.1f us per forward (%.1f Medges/s)" % (k, k, dt * 1e6, m / dt / 1e6))
