"""How much does the stream -> hardware-queue placement matter for batches in flight?  Re-draws the slots' streams from
torch's pool and reports the throughput of each draw."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcnn_keras_amd import synth
from gcnn_keras_amd.engine import SchnetForward

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
b = synth.qm9_like_batch(num_graphs=128, seed=1234)
p = synth.schnet_params(seed=7)
fwd = SchnetForward(p, depth=3, mode="fused", in_flight=k)
fwd.load_batch(b)


def rate(steps=300):
    for i in range(40):
        fwd.replay(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fwd.replay(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


print("GPU_MAX_HW_QUEUES=%s in_flight=%d" % (os.environ.get("GPU_MAX_HW_QUEUES"), k))
for draw in range(10):
    print("draw %d: %.1f us/step   streams %s" % (draw, rate(), [hex(s.stream.cuda_stream)[-5:] for s in fwd._slots]))
    for s in fwd._slots:
        s.stream = torch.cuda.Stream()
        s._stream_ptr = None
