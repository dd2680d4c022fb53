"""Host cost of one replayed launch group (bench.py's timed loop body) - cProfile + plain timing."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from gcnn_keras_amd import synth
from gcnn_keras_amd.engine import SchnetForward

batch = synth.qm9_like_batch(num_graphs=128, seed=1234)
fwd = SchnetForward(synth.schnet_params(seed=7), depth=3, in_flight=4, group=5)
fwd.load_batch(batch)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(4):
        fwd.replay_group(j)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("4 group launches: host %.1f us, until drained %.1f us" % ((t1 - t0) * 1e6, (t2 - t0) * 1e6))
torch.cuda.synchronize()
t0 = time.perf_counter()
fwd.replay_group(0)
torch.cuda.synchronize()
print("one group alone: %.1f us" % ((time.perf_counter() - t0) * 1e6))
pr = cProfile.Profile()
pr.enable()
for j in range(200):
    fwd.replay_group(j)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
