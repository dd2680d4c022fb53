"""Whole-forward drivers of the SchNet hot path (kgcnn/literature/Schnet.py:104-148) on one GPU.

``SchnetForward`` owns resident batches and calls ``Schnet.make_model(...)(inputs)`` on them, in one of two modes:

* ``layers``: the reference's layer graph, one engine primitive per Keras layer (``model(inputs, fused=False)``).
* ``fused``:  the model's own fused route - the same arithmetic in eight kernels replayed from a HIP graph
  (csrc/mp_schnet_*.hip, gcnn_keras_amd/fused.py).

Both go through the C ABI only; there is no CPU path.
"""
import ctypes
import os

import torch

from . import _ffi
from .ragged import RaggedTensor


class _HipTimer:
    """HIP events on torch's current stream via the C ABI (``torch.cuda.Event`` would also see only that stream)."""

    def __init__(self):
        self.start, self.stop = ctypes.c_void_p(), ctypes.c_void_p()
        _ffi.call("mp_event_create", ctypes.byref(self.start))
        _ffi.call("mp_event_create", ctypes.byref(self.stop))

    def time_ms(self, fn, iters):
        fn()
        torch.cuda.synchronize()
        _ffi.call("mp_event_record", self.start, _ffi.stream())
        for _ in range(iters):
            fn()
        _ffi.call("mp_event_record", self.stop, _ffi.stream())
        ms = ctypes.c_float(0.0)
        _ffi.call("mp_event_elapsed_ms", self.start, self.stop, ctypes.byref(ms))
        return ms.value / iters

    def __del__(self):
        try:
            _ffi.call("mp_event_destroy", self.start)
            _ffi.call("mp_event_destroy", self.stop)
        except Exception:
            pass


class SchnetForward:
    """Bench / test harness around ``Schnet.make_model``: owns resident batches and calls the model on them.

    ``mode="fused"`` lets the model take its fused route (``model.fused``, gcnn_keras_amd/fused.py), ``"layers"`` forces
    the reference's layer sequence (``model(inputs, fused=False)``).  ``in_flight``: number of independent input sets
    (each bound to its own batch slot by the model) served round-robin on their own HIP streams - a 128-graph batch
    fills neither the chip (819 edge tiles on 1024 SIMDs, 144 node tiles on 256 CUs) nor a SIMD's matrix pipe (FP32 MFMA
    and the wave's own vector phases exclude each other), so a serving loop that keeps several batches in flight lets
    the kernels of different batches share CUs: ``replay(i)`` is ``model(inputs[i % in_flight])`` on stream
    ``i % in_flight``."""

    def __init__(self, params, depth=3, mode="auto", units=128, bins=20, gauss_args=None, in_flight=1, group=1):
        from .literature import Schnet
        if not torch.cuda.is_available():
            raise _ffi.EngineError("SchnetForward needs an MI355X (no CPU fallback)")
        self.params = params
        self.depth = depth
        self.units = units
        self.gauss_args = gauss_args or {"bins": bins, "distance": 4, "offset": 0.0, "sigma": 0.4}
        self.model = Schnet.make_model(depth=depth, gauss_args=self.gauss_args)
        self.model.set_weights(list(params.values()))
        if mode == "auto":
            mode = "fused" if self.model.fused is not None else "layers"
        if mode == "fused" and self.model.fused is None:
            raise ValueError("this SchNet configuration does not fit the fused kernels")
        self.mode = mode
        self.num_launches = None
        self._batch = None
        self._inputs = []
        self._streams = []
        self._slots = []
        self.placement = None
        self.in_flight = max(1, int(in_flight)) if mode == "fused" else 1
        # launch groups: `group` independent batches served by ONE launch sequence (route.call_group: concatenated on the
        # device), `in_flight` such groups in flight on their own streams
        self.group = max(1, int(group)) if mode == "fused" else 1
        self._group_inputs = []
        # several forwards in flight use the same 4-wave cfconv build as a lone forward (flag bit 4 - the 8-wave build,
        # two waves per SIMD on one LDS image - measured equal within run-to-run spread: 634 vs 621 M edges/s, and it
        # costs a lone forward 12 us); MPENGINE_INFLIGHT_CFCONV_FLAGS overrides for experiments
        if mode == "fused" and self.in_flight > 1 and os.environ.get("MPENGINE_INFLIGHT_CFCONV_FLAGS"):
            self.model.fused.cfconv_flags = int(os.environ["MPENGINE_INFLIGHT_CFCONV_FLAGS"])
        # launch groups in flight: the node chains of a union launch (hundreds of tiles) run on half the CUs - flag bit 9 of
        # the node entry points, "several launch sequences in flight" - so that the other sequences' kernels find CUs and
        # every workgroup's weight slices serve twice the tiles (+3 %); single batches (144 tiles) are not affected
        if mode == "fused" and self.in_flight > 1 and os.environ.get("MPENGINE_INFLIGHT_NODE_HALF", "1") != "0":
            self.model.fused.cfconv_flags |= 512
        if mode == "fused":
            self.model.fused.max_slots = max(self.model.fused.max_slots, self.in_flight)

    @property
    def _fused(self):
        return self._slots[0] if self._slots else None

    # ------------------------------------------------------------------------------------------------ batch
    def load_batch(self, batch):
        """Host arrays -> HBM (outside the timed region: inputs are resident when the clock starts)."""
        dev = "cuda"
        self._batch = {
            "z": torch.from_numpy(batch["node_number"]).to(dev),
            "xyz": torch.from_numpy(batch["node_coordinates"]).to(dev),
            "idx": torch.from_numpy(batch["edge_indices"]).to(dev),
            "ns": torch.from_numpy(batch["node_splits"]).to(dev),
            "es": torch.from_numpy(batch["edge_splits"]).to(dev),
            "ns_host": batch["node_splits"], "es_host": batch["edge_splits"],
        }
        self.N = int(batch["node_splits"][-1])
        self.M = int(batch["edge_splits"][-1])
        self.G = len(batch["node_splits"]) - 1
        if self.mode != "fused":
            return
        self.model.fused.release()
        self._inputs, self._streams, self._slots = [], [], []
        for k in range(self.in_flight):
            # every slot owns a copy of the inputs, as it would hold a different batch in production
            own = self._batch if k == 0 else {key: (v.clone() if torch.is_tensor(v) else v)
                                              for key, v in self._batch.items()}
            self._inputs.append(self._ragged_inputs(own))
            self._streams.append(torch.cuda.Stream())
        torch.cuda.synchronize()
        for k in range(self.in_flight):
            with torch.cuda.stream(self._streams[k]):
                self.model(self._inputs[k])   # binds the batch slot (index pass, buffers), direct launch
                self.model(self._inputs[k])   # captures the slot's graph now, outside any timed region
            self._slots.append(self.model.fused.slot_of(self._inputs[k]))
        if self.group > 1:
            self.model.fused.max_groups = max(self.model.fused.max_groups, self.in_flight)
            self._group_inputs = []
            for k in range(self.in_flight):
                members = []
                for _ in range(self.group):        # every member is its own set of tensors, as different batches would be
                    own = {key: (v.clone() if torch.is_tensor(v) else v) for key, v in self._batch.items()}
                    members.append(self._ragged_inputs(own))
                self._group_inputs.append(members)
            torch.cuda.synchronize()
            for k in range(self.in_flight):
                with torch.cuda.stream(self._streams[k]):
                    self.model.fused.call_group(self._group_inputs[k])    # bind + direct launch
                    self.model.fused.call_group(self._group_inputs[k])    # capture, outside any timed region
                    self.model.fused.call_group(self._group_inputs[k])
                    self.model.fused.call_group(self._group_inputs[k])    # (the result ring's further executables)
        torch.cuda.synchronize()
        if self.in_flight > 1:
            self._place_streams()

    def _place_streams(self, draws=12, steps=150):
        """How well forwards in flight overlap depends on which hardware queues the streams land on (the ROCm runtime
        multiplexes streams onto a few queues; a stream sharing a queue with another serialises behind it: measured 48 vs
        64 us per step for different draws of four streams from torch's pool).  This draws the streams a few times,
        measures ~150 forwards twice per draw - a draw is scored by the SLOWER of its two runs, so that one lucky run does
        not select a placement that is slow most of the time - and keeps the best draw.  Runs once per ``load_batch``,
        outside any timed region."""
        import time

        one = self.replay_group if self.group > 1 else self.replay
        if self.group > 1:
            steps = max(steps // self.group, 4 * self.in_flight)

        def rate():
            for i in range(2 * self.in_flight):
                one(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                one(i)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        best, best_streams, seen = None, None, []
        for _ in range(draws):
            t = max(rate(), rate())
            seen.append(t / steps)
            if best is None or t < best:
                best, best_streams = t, list(self._streams)
            self._streams = [torch.cuda.Stream() for _ in range(self.in_flight)]
        self._streams = best_streams
        seen.sort()
        # what a caller who simply takes four streams from torch's pool gets is a random draw: the median beside the best
        per = 1e6 / self.group            # a group launch is `group` steps
        self.placement = {"draws": draws, "steps_per_draw": steps * self.group, "best_us_per_step": seen[0] * per,
                          "median_us_per_step": seen[len(seen) // 2] * per, "worst_us_per_step": seen[-1] * per}
        torch.cuda.synchronize()

    @staticmethod
    def _ragged_inputs(b):
        def rag(v, s, sh):
            r = RaggedTensor(v, s)
            r._splits_host = sh  # partition sizes are batch metadata from the host-side loader
            return r

        return [rag(b["z"], b["ns"], b["ns_host"]), rag(b["xyz"], b["ns"], b["ns_host"]),
                rag(b["idx"], b["es"], b["es_host"])]

    def _fresh_inputs(self):
        """New ragged wrappers every step, so nothing cached on them (index plan, CSR) leaks across steps."""
        return self._ragged_inputs(self._batch)

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, step=0):
        """``model(inputs)`` on the caller's current stream (slot ``step % in_flight``'s inputs)."""
        if self.mode == "fused":
            return self.model(self._inputs[step % self.in_flight])
        before = _ffi.launch_count()
        out = self.model(self._fresh_inputs(), fused=False)
        self.num_launches = _ffi.launch_count() - before
        return out

    def replay(self, step=0, restore_stream=True):
        """Fused mode: ``model(inputs[k])`` on stream ``k = step % in_flight`` (a re-bound batch: the model replays the
        slot's captured graph - one hipGraphLaunch - and copies the (G,1) result out); layers mode: same as ``forward``."""
        if self.mode == "fused":
            k = step % self.in_flight
            if restore_stream:
                with torch.cuda.stream(self._streams[k]):
                    return self.model(self._inputs[k])
            # a serving loop: set_stream instead of the context manager (two stream switches and a device guard per call,
            # ~8 us of a ~30 us call); torch's current stream STAYS slot k's stream - the caller switches back when it is
            # done (bench.py: torch.cuda.set_stream(default) before its closing barrier)
            torch.cuda.set_stream(self._streams[k])
            return self.model(self._inputs[k])
        return self.forward(step)

    def replay_group(self, j=0):
        """Launch group ``j % in_flight`` on its stream: ``group`` forwards from one launch sequence (``route.call_group``);
        torch's current stream stays that stream (a serving loop: the caller switches back when it is done)."""
        k = j % self.in_flight
        torch.cuda.set_stream(self._streams[k])
        return self.model.fused.call_group(self._group_inputs[k])

    @property
    def stream(self):
        """The HIP stream slot 0 is served on."""
        return self._streams[0] if self._streams else torch.cuda.current_stream()

    def stream_of(self, step):
        """Stream of the slot that serves ``step``."""
        return self._streams[step % self.in_flight] if self._streams else torch.cuda.current_stream()

    def check_flags(self):
        if self.model.fused is not None:
            self.model.fused.check_flags()

    # ------------------------------------------------------------------------------------------------ roofline
    def roofline(self, hbm_peak_gbs, mfma_peak_tf, iters=50):
        """Dominant kernel of the forward, timed live with HIP events on the stream it is launched on."""
        if self.mode == "fused":
            self.num_launches = self._fused.num_launches
            if self.group > 1:      # the kernel as it runs in the timed loop: on the union of a launch group's members
                route = self.model.fused
                grp = route._groups.get(tuple(route._key(*x) for x in self._group_inputs[0]))
                if grp is not None:
                    roof = grp.slot.roofline(hbm_peak_gbs, mfma_peak_tf, iters)
                    roof["kernel"] += " on the union of %d batches (%d edges, %d edge tiles)" % (
                        self.group, grp.slot.M, (grp.slot.M + 31) // 32)
                    return roof
            return self._fused.roofline(hbm_peak_gbs, mfma_peak_tf, iters)
        # layers mode: the per-edge filter GEMM (M,128)x(128,128) of SchNetCFconv.lay_dense2 dominates
        m, f = self.M, self.units
        x = torch.randn(m, f, device="cuda")
        w = torch.randn(f, f, device="cuda") * 0.1
        b = torch.zeros(f, device="cuda")
        out = torch.empty(m, f, device="cuda")

        def launch():
            _ffi.call("mp_dense_f32", _ffi.ptr(x), m, f, _ffi.ptr(w), _ffi.ptr(b), f, 0, 0.0, _ffi.ptr(out),
                      _ffi.stream())

        ms = _HipTimer().time_ms(launch, iters)
        flops = 2.0 * m * f * f
        achieved = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "dense_mfma_kernel (edge filter GEMM (M,128)x(128,128))",
                "achieved": achieved, "peak": mfma_peak_tf, "unit": "TFLOP/s", "frac": achieved / mfma_peak_tf,
                "traffic": None, "avg_launch_us": ms * 1e3, "algorithmic_flops_per_launch": flops}


class GraphedModel:
    """Replay a layer-path model (any ``make_model`` result, or an ``EnergyForceModel``) from one HIP graph.

    The layer path issues ~100 engine calls per forward (several hundred with the reverse pass of an
    ``EnergyForceModel``); at QM9/MD17 batch sizes the host cannot keep the GPU busy.  ``GraphedModel(model, inputs)``
    runs the call eagerly (which builds and caches the index plans on the given ragged inputs, so no host
    synchronisation is left), then captures the same call sequence - every launch goes through the C ABI on the capture
    stream, intermediate buffers come from the graph's private pool - and ``__call__()`` replays it.  With
    ``grad=True`` (default for an ``EnergyForceModel``) the capture includes the reverse pass that produces the forces.
    The graph is bound to the input buffers and to the batch's index structure: refresh feature / coordinate *values*
    in place (``inputs[i].values.copy_(...)``) between replays; a new edge list needs a new graph.

    A captured fused route writes into the work buffers of the route's batch slot and reads the route's packed weight
    images; the route's own slot table is least-recently-used with a small capacity, so this object holds the slot (and
    the images) itself: binding more batches than ``max_slots``, or ``route.release()``, cannot free memory the graph
    still uses.  ``__call__`` first lets the route refresh its derived weight layouts (``set_weights`` after the capture
    is picked up: the images are re-filled in place) and raises if weight tensors were REPLACED by other objects - the
    graph holds the old addresses, capture again.
    """

    def __init__(self, model, inputs, grad=None):
        if not torch.cuda.is_available():
            raise _ffi.EngineError("GraphedModel needs an MI355X (no CPU fallback)")
        if grad is None:
            grad = hasattr(model, "energy_model")
        self.model, self.inputs, self.grad = model, inputs, bool(grad)
        self._routes, self._pinned = [], []
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        # Models with a fused route replay their own HIP graph per bound batch; launching that graph into a capturing
        # stream would nest graphs, so for the capture the route issues its kernels directly (they land in THIS graph).
        routes = [r for r in (getattr(model, "fused", None), getattr(getattr(model, "energy_model", None), "fused", None))
                  if r is not None and hasattr(r, "mode")]
        saved = [r.mode for r in routes]
        for r in routes:
            r.mode = "eager"
        auto = [m for m in (model, getattr(model, "energy_model", None)) if getattr(m, "auto_graph", False)]
        for m in auto:          # a model that graphs itself must not replay its own graph inside this capture
            m.auto_graph = False
        try:
            with torch.cuda.stream(self.stream), torch.set_grad_enabled(self.grad):
                for _ in range(3):
                    model(inputs)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.set_grad_enabled(self.grad), torch.cuda.graph(self.graph, stream=self.stream):
                self.output = model(inputs)
            torch.cuda.synchronize()
            for r in routes:   # keep what the captured kernels address alive, whatever the route's slot table does later
                for table in (getattr(r, "_slots", None), getattr(r, "_gslots", None)):
                    if table:
                        self._pinned.append(next(reversed(table.values())))      # most recently used = this capture's
                self._pinned.extend(getattr(r, a, None) for a in ("_packed", "_images", "_grad_images", "_p"))
                self._routes.append((r, self._weight_objects(r)))
        finally:
            for r, mode in zip(routes, saved):
                r.mode = mode
            for m in auto:
                m.auto_graph = True

    @staticmethod
    def _weight_objects(route):
        key = getattr(route, "_wkey", None)
        return None if key is None else tuple(k[:2] for k in key)    # (id, storage address) per weight tensor

    def __call__(self):
        for r, objects in self._routes:
            r._sync_weights()
            if self._weight_objects(r) != objects:
                raise _ffi.EngineError("GraphedModel: weight tensors of the model were replaced after the capture (the graph "
                                       "holds the old addresses); build a new GraphedModel")
        self.graph.replay()
        return self.output


class GraphedModelPool:
    """Several ``GraphedModel`` replicas - one per batch slot, each with its own inputs, private graph memory and HIP
    stream - so that forwards (or energy + force passes) of different batches overlap on the GPU, for ANY layer-path
    model.  ``replay(i)`` launches slot ``i % in_flight`` on that slot's stream and returns its (static) output; call
    ``wait(i)`` (or synchronise) before reading it from another stream."""

    def __init__(self, model, inputs_per_slot, grad=None):
        self.slots = [GraphedModel(model, inputs, grad=grad) for inputs in inputs_per_slot]
        self.in_flight = len(self.slots)
        self._events = [None] * self.in_flight

    def replay(self, step=0):
        k = step % self.in_flight
        slot = self.slots[k]
        with torch.cuda.stream(slot.stream):
            slot()
            ev = torch.cuda.Event()
            ev.record()
        self._events[k] = ev
        return slot.output

    def wait(self, step=0):
        ev = self._events[step % self.in_flight]
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
