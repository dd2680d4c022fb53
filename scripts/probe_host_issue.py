"""Is the in-flight loop of bench.py bound by the host's call rate or by the GPU?  Time to ISSUE K model(inputs) calls
(before the final synchronisation) against the time until the GPU has finished them."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

from gcnn_keras_amd import synth
from gcnn_keras_amd.engine import SchnetForward

b = synth.qm9_like_batch(num_graphs=128, seed=1234)
fwd = SchnetForward(synth.schnet_params(seed=7), depth=3, in_flight=4)
fwd.load_batch(b)
for k in (20, 200, 2000):
    for _ in range(3):
        for i in range(5):
            fwd.replay(i, restore_stream=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            fwd.replay(i, restore_stream=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("K=%4d  issue %.1f us/call   total %.1f us/step   (issue %.0f us, total %.0f us)"
              % (k, (t1 - t0) / k * 1e6, (t2 - t0) / k * 1e6, (t1 - t0) * 1e6, (t2 - t0) * 1e6), flush=True)
