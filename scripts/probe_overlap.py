"""Which kernels of the fused SchNet forward overlap when several batches are in flight?  Four batch slots (config 2),
each with a captured graph of a SUBSET of the forward's launches (numerics are irrelevant here), replayed round-robin on
four streams; time per graph launch with 1, 2 and 4 slots in flight.

    python scripts/probe_overlap.py
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

from gcnn_keras_amd import _ffi, synth
from gcnn_keras_amd.fused import FusedSchnet


class Subset(FusedSchnet):
    only = "all"

    def _launch_all(self, out=None):
        if self.only == "all":
            return FusedSchnet._launch_all(self, out)
        p, w = self.p, self.node_images
        for i in range(self.depth):
            pre = "interaction%d/" % i
            if self.only in ("cfconv", "cfconv+node"):
                self._cfconv(i, self.agg)
            if self.only in ("node", "cfconv+node"):
                _ffi.call("mp_schnet_node_update_f32", _ffi.ptr(self.agg), self.N, _ffi.ptr(w[pre + "dense2/kernel"]),
                          _ffi.ptr(p.get(pre + "dense2/bias")), _ffi.ptr(w[pre + "dense3/kernel"]),
                          _ffi.ptr(p.get(pre + "dense3/bias")), _ffi.ptr(self.n),
                          _ffi.ptr(w["interaction%d/dense1/kernel" % ((i + 1) % self.depth)]), _ffi.ptr(self.x),
                          self.node_flags, _ffi.stream())
        if self.only == "small":   # the two latency-floor kernels
            wo0, bo0, wo1, bo1 = self._head()
            for _ in range(3):
                _ffi.call("mp_schnet_readout_f32", _ffi.ptr(self.h), _ffi.ptr(self._b["ns"]), self.G, _ffi.ptr(wo0),
                          _ffi.ptr(bo0), _ffi.ptr(wo1), _ffi.ptr(bo1), _ffi.ptr(self.out), _ffi.stream())


def main():
    b = synth.qm9_like_batch(num_graphs=128, seed=1234)
    params = synth.schnet_params(seed=7)
    res = {}
    flag_sets = [int(a) for a in sys.argv[1:]] or [0]
    for only, cf in [(o, f) for f in flag_sets for o in (("all", "cfconv", "node", "cfconv+node", "small") if len(flag_sets) == 1
                                                      else ("all", "cfconv"))]:
        slots = []
        for k in range(4):
            s = Subset(params, depth=3, cfconv_flags=cf)
            s.only = "all"
            batch = {"z": torch.from_numpy(b["node_number"]).cuda(), "xyz": torch.from_numpy(b["node_coordinates"]).cuda(),
                     "idx": torch.from_numpy(b["edge_indices"]).cuda(), "ns": torch.from_numpy(b["node_splits"]).cuda(),
                     "es": torch.from_numpy(b["edge_splits"]).cuda(), "ns_host": b["node_splits"]}
            s.bind(batch, int(b["node_splits"][-1]), int(b["edge_splits"][-1]), 128)
            s._launch_all()          # fills every buffer once
            torch.cuda.synchronize()
            s.only = only
            s.graph = None
            s.forward()              # captures the subset
            slots.append(s)
        torch.cuda.synchronize()
        row = {}
        for nfl in (1, 2, 4):
            best = None
            for _ in range(3):
                for i in range(40):
                    slots[i % nfl].replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(400):
                    slots[i % nfl].replay()
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / 400 * 1e6
                best = t if best is None else min(best, t)
            row["us_per_graph_%d_in_flight" % nfl] = round(best, 2)
        res["%s/flags%d" % (only, cf)] = row
        print(only, "cfconv_flags", cf, row, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
