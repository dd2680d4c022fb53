"""Is the in-flight loop host-bound?  Times the bare hipGraphLaunch call and compares one launching thread with one
thread per batch slot (ctypes drops the GIL during the call)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.engine import SchnetForward

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b = synth.qm9_like_batch(num_graphs=128, seed=1234)
p = synth.schnet_params(seed=7)
fwd = SchnetForward(p, depth=3, mode="fused", in_flight=k)
fwd.load_batch(b)
m = int(b["edge_splits"][-1])
steps = 800

for i in range(40):
    fwd.replay(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
one = (lambda i: fwd._slots[i % k].launch_direct()) if os.environ.get("MP_MODE") == "direct" else fwd.replay
for i in range(steps):
    one(i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(os.environ.get("MP_MODE", "graph"), "one thread : host issue %.1f us/step, completed %.1f us/step" % (t_host / steps * 1e6, t_all / steps * 1e6))


MODE = os.environ.get("MP_MODE", "graph")


def worker(j, n):
    slot = fwd._slots[j]
    call = slot.launch_direct if MODE == "direct" else slot.replay
    for _ in range(n):
        call()


for rep in range(2):
    ths = [threading.Thread(target=worker, args=(j, steps // k)) for j in range(k)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%d threads  : host issue %.1f us/step, completed %.1f us/step  (%.1f Medges/s)" % (
        k, t_host / steps * 1e6, t_all / steps * 1e6, m * steps / t_all / 1e6))
