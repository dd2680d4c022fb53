#!/usr/bin/env python
"""bench.py - edges/sec of a SchNet forward on QM9-shaped synthetic batches (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one forward of ``gcnn_keras_amd.literature.Schnet.make_model`` (the mirror of kgcnn.literature.Schnet.make_model:
F=128, depth 3, Gauss(20, 4.0, 0.4), sum pooling) on ONE 128-graph batch whose raw API inputs (float node numbers, float32
coordinates, int64 (M,2) sample indices, int64 row_splits) are resident in HBM when the clock starts; index preparation
(shift, receiver/sender split), geometry, basis expansion, every interaction block, readout and output MLP are inside it,
and every step's result is a fresh tensor.  The model routes such a forward through its fused HIP kernels (8 launches,
replayed from a HIP graph for a re-bound batch; gcnn_keras_amd/fused.py).

Workloads (``--workload``; default ``config2`` at N = 1 and ``config4`` at N > 1):

* ``config2``  BASELINE config 2, the configuration the metric is quoted on: 128 QM9-shaped graphs (seed 1234 + rank:
  N=2301, M=26190 on rank 0).  One 128-graph forward cannot fill an MI355X (819 edge tiles for 1024 SIMDs, 144 node
  tiles for 256 CUs, eight kernel boundaries of ~4 us), so the loop serves independent batches - own tensors each - in
  LAUNCH GROUPS: ``--group`` (default 5) batches are concatenated on the device by one kernel and run by one launch sequence
  (``model.fused.call_group``; every member gets the rows a forward of its own gives), and ``--in-flight`` (default 4) such
  groups overlap on their own HIP streams.  Exactly K forwards run inside the timed region: K // group group launches + K %
  group single ``model(inputs)`` calls.  ``--group 1`` is round 2's mode (every batch its own ``model(inputs)`` call, four in
  flight); ``single_forward_latency_ms`` reports the latency of a lone ``model(inputs)`` call beside the throughput, and
  ``stream_fresh_batches`` the rate for batches that are seen ONCE (bind + direct launch, no replay).  With N > 1
  (``--workload config2``) every rank owns its own 128-graph batches and one all-gather of the (G,1) predictions follows
  every forward: weak scaling.
* ``config4``  BASELINE config 4: 100 000 QM9-shaped molecules (seed 3456) cut into N contiguous shards balanced by edge
  count (gcnn_keras_amd/sharding.py); rank r builds the edge lists of shard r on its GPU (the engine's SetRange, the
  reference's rule), runs one forward per step on it - no exchange during the forward - and ONE RCCL all-gather returns
  the (100000,1) predictions in graph order.  Total work is fixed as N grows: strong scaling.  At N = 1 the default run
  appends this workload's single-GPU rate as ``config4_single_gpu`` (the N = 1 point of that curve).

Prints ONE JSON line (rank 0) with the contract keys plus ``roofline`` (dominant kernel, measured live with HIP events)
and ``cpu_baseline`` (CPU restatements of the reference's unfused op sequence - C/OpenMP port and torch-CPU - timed on
the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md chip table)
FP32_MFMA_PEAK_TF = 157.3  # dense FP32 matrix peak (same table)
DEPTH = 3


def schnet_algorithmic_bytes(n, m, g, f=128, b=20, d=3):
    """bytes = [16N + 16M + 4MB + 4NF] + D[8NF + 4MB + 16M] + [4NF + 4G]."""
    return (16 * n + 16 * m + 4 * m * b + 4 * n * f) + d * (8 * n * f + 4 * m * b + 16 * m) + (4 * n * f + 4 * g)


def schnet_flops(n, m, g, f=128, b=20, d=3):
    """flops = M D [2(BF + F^2) + 2F] + N [2*64F + D 6F^2 + 2(F^2 + 64F)] + G 2(64^2 + 64)."""
    return (m * d * (2 * (b * f + f * f) + 2 * f) + n * (2 * 64 * f + d * 6 * f * f + 2 * (f * f + 64 * f))
            + g * 2 * (64 * 64 + 64))


def _timed(run, budget_s, max_runs=200, warmup=1):
    for _ in range(warmup):
        run()
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < max_runs:
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), len(times)


def _protocol(run, forwards=50, warmup=5, cap_s=20.0):
    """warm-up 5, then 50 forwards of the same batch; median and p10 / p90 of the per-forward
    times.  ``cap_s`` bounds the sample on a slow host (the count actually run is reported)."""
    for _ in range(warmup):
        run()
    times = []
    t_end = time.perf_counter() + cap_s
    while len(times) < forwards and (time.perf_counter() < t_end or len(times) < 5):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    t = np.asarray(times)
    return {"median": float(np.median(t)), "p10": float(np.percentile(t, 10)), "p90": float(np.percentile(t, 90)),
            "forwards": int(t.size)}


def _best_threads(run, set_threads, candidates, probe_s=1.0):
    """A 128-graph batch does not scale to every core of a 256-thread host: probe a few thread counts briefly and keep
    the fastest (what a user tuning the CPU run would do)."""
    best = None
    for n in candidates:
        set_threads(n)
        med, _ = _timed(run, probe_s, max_runs=20)
        if best is None or med < best[0]:
            best = (med, n)
    set_threads(best[1])
    return best[1]


def cpu_baseline(batch, params, depth):
    """CPU restatements of the reference's unfused op sequence on this host's cores: (i) the
    C/OpenMP port oracle/mp_oracle.c, (ii) torch-CPU (index_select / index_add_ / addmm), each at the thread count that
    is fastest on this host (a one-second probe per candidate), then the protocol's sample: warm-up 5 + 50 forwards of the
    same batch, median with p10 / p90.  ``value`` is the faster of the two medians; both legs are listed."""
    from oracle import c_oracle, torch_oracle
    import torch
    m = int(batch["edge_splits"][-1])
    g = len(batch["node_splits"]) - 1
    ncpu = os.cpu_count() or 1
    candidates = sorted({n for n in (8, 16, 32, 64, 128, ncpu // 2, ncpu) if 1 <= n <= ncpu})
    legs = {}
    if c_oracle.available():
        run = lambda: c_oracle.schnet_forward(params, batch["node_number"], batch["node_coordinates"],
                                              batch["edge_indices"], batch["node_splits"], batch["edge_splits"],
                                              depth=depth)
        threads = _best_threads(run, c_oracle.set_num_threads, candidates)
        st = _protocol(run)
        legs["c_openmp"] = {"value": m / st["median"], "median_ms": st["median"] * 1e3, "p10_ms": st["p10"] * 1e3,
                            "p90_ms": st["p90"] * 1e3, "forwards": st["forwards"], "threads": threads,
                            "what": "C/OpenMP port oracle/mp_oracle.c"}
    tp, tb = torch_oracle.to_torch(params), torch_oracle.prepare(batch)
    run = lambda: torch_oracle.schnet_forward(tp, tb, depth=depth)
    threads = _best_threads(run, torch.set_num_threads, candidates)
    st = _protocol(run)
    legs["torch_cpu"] = {"value": m / st["median"], "median_ms": st["median"] * 1e3, "p10_ms": st["p10"] * 1e3,
                         "p90_ms": st["p90"] * 1e3, "forwards": st["forwards"], "threads": threads,
                         "what": "torch-CPU restatement oracle/torch_oracle.py (index_select / index_add_ / addmm)"}
    best = max(legs, key=lambda k: legs[k]["value"])
    return {"value": legs[best]["value"], "unit": "edges/s", "cores": int(legs[best]["threads"]), "kind": "port",
            "sample": "warm-up 5 + %d forwards of the same %d-graph batch (median %.1f ms, p10 %.1f, p90 %.1f) with the %s "
                      "on %d threads (fastest of %s threads in a 1-s probe each; host has %d logical cores)" % (
                          legs[best]["forwards"], g, legs[best]["median_ms"], legs[best]["p10_ms"], legs[best]["p90_ms"],
                          legs[best]["what"], legs[best]["threads"], candidates, ncpu),
            "legs": legs}


def _kernel_entries(kernels, kernel_name):
    """Entries of a counter summary that belong to ``kernel_name`` ("cfconv_fused_kernel<8,gauss> ..."): the build named there
    first (a launch-group run holds the 8-wave union launches next to 4-wave single-batch launches of the same kernel), any
    build of the kernel otherwise."""
    exact = [(k, v) for k, v in kernels.items() if k.split(",")[0] in kernel_name]
    return exact or [(k, v) for k, v in kernels.items() if k.split("<")[0] in kernel_name]


def pmc_traffic(kernel_name, graphs):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*pmc*.json):
    FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command; FETCH doubled as the microarch guide
    prescribes for gfx950.  None when no pass exists for this workload size."""
    import glob
    want = {128: "config 2 (", 640: "config 2 groups", 12500: "12500 graphs"}.get(graphs)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")), reverse=True):
        try:
            for entry in json.load(open(path)):
                if want and entry["label"].startswith(want):
                    for k, v in _kernel_entries(entry["kernels"], kernel_name):
                        if True:
                            raw_f, raw_w = v["FETCH_SIZE_KB_median"] * 1024, v["WRITE_SIZE_KB_median"] * 1024
                            return {"traffic": 2 * raw_f + raw_w,
                                    "traffic_detail": {"source": os.path.basename(path), "fetch_bytes_raw": raw_f,
                                                       "write_bytes": raw_w, "fetch_correction": "x2 (gfx950)"}}
        except Exception:
            continue
    return {"traffic": None}


def pmc_matrix_pipe(kernel_name, graphs, avg_launch_us):
    """Matrix-pipe occupancy of the dominant kernel from the committed SQ counter pass (profiles/r*_pmc_mfma.json,
    scripts/run_pmc_mfma.sh): SQ_VALU_MFMA_BUSY_CYCLES per launch (32 per v_mfma_f32_32x32x16_bf16) over the cycles of
    all 1024 SIMDs - once with the kernel's cycles from GRBM_GUI_ACTIVE / 8 (the chip's own clock under load; reads high on
    dispatches shorter than ~0.3 ms, so the share reads LOW there) and once with this run's measured launch time at the
    2.4 GHz the roofline peak is quoted at.  Empty when no pass exists for this workload size."""
    import glob
    want = {128: "config 2 (", 640: "config 2 groups", 12500: "12500 graphs"}.get(graphs)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_mfma.json")), reverse=True):
        try:
            for entry in json.load(open(path)):
                if want and entry["label"].startswith(want):
                    for k, v in _kernel_entries(entry["kernels"], kernel_name):
                        if v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                            busy = v["SQ_VALU_MFMA_BUSY_CYCLES"]
                            out = {"source": os.path.basename(path), "SQ_VALU_MFMA_BUSY_CYCLES": busy,
                                   "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"),
                                   "kernel_cycles_grbm": v.get("kernel_cycles"),
                                   "matrix_pipe_busy_grbm_clock": v.get("matrix_pipe_busy")}
                            if avg_launch_us:
                                out["matrix_pipe_busy_at_2p4GHz"] = busy / (1024.0 * avg_launch_us * 1e-6 * 2.4e9)
                            return {"matrix_pipe_pmc": out}
        except Exception:
            continue
    return {}


class _Dist:
    """torch.distributed plumbing of one rank (RCCL = backend "nccl"; gloo only rehearses N > 1 on a one-GPU box)."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, self.world, args.gpus))
        ndev = torch.cuda.device_count()
        self.device_index = self.local_rank % max(ndev, 1)  # one rank per GPU; wraps only in one-GPU rehearsals
        torch.cuda.set_device(self.device_index)
        self.backend = args.backend
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.device_index))
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, value, op):
        t = self.torch.tensor([value], dtype=self.torch.float64,
                              device="cuda" if (self.backend == "nccl" or self.world == 1) else "cpu")
        if self.dist is not None:
            self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def _time_steps(d, step, warmup, steps):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks."""
    torch = d.torch
    for i in range(warmup):
        step(i)
    d.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.set_stream(torch.cuda.default_stream())
    d.barrier()
    return d.reduce(time.perf_counter() - t0, "MAX")


# ---------------------------------------------------------------------------------------------------- config 2
def run_config2(args, d):
    torch = d.torch
    from gcnn_keras_amd import synth
    from gcnn_keras_amd.engine import SchnetForward
    world = d.world
    batch = synth.qm9_like_batch(num_graphs=args.graphs, seed=1234 + d.rank)
    params = synth.schnet_params(seed=7)  # Keras defaults: glorot_uniform kernels, zero biases, U(-0.05, 0.05) embedding
    n_nodes, n_edges, n_graphs = int(batch["node_splits"][-1]), int(batch["edge_splits"][-1]), args.graphs

    # default: launch groups of 5 batches, 4 groups in flight (sweep in DESIGN section 5); an explicit --in-flight without
    # --group keeps single batches
    group = args.group if args.group else (5 if args.in_flight is None else 1)
    if world > 1:
        group = 1                                  # N > 1 (weak scaling rehearsal): one forward + one all-gather per step
    in_flight = args.in_flight if args.in_flight else 4
    fwd = SchnetForward(params, depth=DEPTH, mode=args.mode, in_flight=in_flight, group=group)
    fwd.load_batch(batch)
    slots, group = fwd.in_flight, fwd.group
    gathered = [torch.empty((world * n_graphs, 1), dtype=torch.float32, device="cuda") for _ in range(slots)] \
        if world > 1 else None
    host_parts = [torch.empty((n_graphs, 1)) for _ in range(world)] if (world > 1 and d.backend == "gloo") else None

    def step(i):
        # one model(inputs) call on the batch's stream; with RCCL the all-gather is ordered after it on that stream
        # (the process group serialises the collectives of different slots in issue order)
        if world == 1:
            return fwd.replay(i, restore_stream=False)
        with torch.cuda.stream(fwd.stream_of(i)):
            out = fwd.forward(i)
            if host_parts is None:
                d.dist.all_gather_into_tensor(gathered[i % slots], out)
            else:
                d.dist.all_gather(host_parts, out.cpu())                # rehearsal only
        return out

    def run_steps(n):
        # exactly n forwards of 128-graph batches: n // group launch groups (`group` batches concatenated on the device and
        # served by one launch sequence, route.call_group), the remainder as single forwards
        full, rem = divmod(n, group)
        for j in range(full):
            fwd.replay_group(j)
        for r in range(rem):
            fwd.replay(r, restore_stream=False)

    if group > 1:
        run_steps(args.warmup)
        d.barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.set_stream(torch.cuda.default_stream())
        d.barrier()
        elapsed = d.reduce(time.perf_counter() - t0, "MAX")
    else:
        elapsed = _time_steps(d, step, args.warmup, args.steps)
    total_edges = d.reduce(float(n_edges), "SUM")
    fwd.check_flags()
    # latency of ONE model(inputs) call with nothing else on the GPU (batch 0 alone), beside the throughput
    for _ in range(10):
        fwd.replay(0, restore_stream=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(100):
        fwd.replay(0, restore_stream=False)
    torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) / 100 * 1e3
    torch.cuda.set_stream(torch.cuda.default_stream())
    roof = fwd.roofline(HBM_PEAK_GBS, FP32_MFMA_PEAK_TF)  # dominant kernel, timed with HIP events on its stream
    roof.update(pmc_traffic(roof.get("kernel", ""), n_graphs * group))
    roof.update(pmc_matrix_pipe(roof.get("kernel", ""), n_graphs * group, roof.get("avg_launch_us")))
    roof["measured"] = ("HIP events on the kernel's stream around back-to-back launches of the kernel alone on the GPU "
                        "(50 launches per HIP-graph replay, so that the host's call rate is not what is timed), after "
                        "the timed region; rocprofv3 --kernel-trace --stats of `bench.py --in-flight 1` (profiles/) "
                        "gives the same average for the kernel inside the forward")
    if d.rank != 0:
        return None
    ms_per_step = elapsed / args.steps * 1e3
    fwd_bytes = schnet_algorithmic_bytes(n_nodes, n_edges, n_graphs, d=DEPTH)
    fwd_flops = schnet_flops(n_nodes, n_edges, n_graphs, d=DEPTH)
    line = {
        "metric": "edges/sec (SchNet fwd, QM9-shape batch)", "value": total_edges * args.steps / elapsed,
        "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "precision": "float32 inputs, weights and results; both filter GEMMs of the cfconv kernel (K=21 and K=128) are "
                     "evaluated on the bf16 matrix pipe as an exact FP32 emulation (3 bf16 pieces per operand, 6 products, "
                     "FP32 accumulate; error equal to the FP32 MFMA chain's: scripts/probes/bf16x3_probe.hip, "
                     "tests/test_gpu_fused.py)",
        "config": {"workload": "BASELINE config 2: Schnet.make_model forward (F=128, depth 3, Gauss 20) on %d QM9-shaped "
                               "graphs per GPU, N=%d nodes, M=%d directed edges on rank 0"
                               % (n_graphs, n_nodes, n_edges),
                   "graphs_per_gpu": n_graphs, "nodes": n_nodes, "edges": n_edges, "mode": fwd.mode,
                   "in_flight": slots, "batches_per_launch_group": group,
                   "api": "gcnn_keras_amd.literature.Schnet.make_model(depth=3)(inputs)" if group == 1 else
                          "Schnet.make_model(depth=3).fused.call_group([inputs_1, ..., inputs_%d])" % group,
                   "step": ("one model(inputs) call on one batch; %d independent batches (own tensors, batch slot and HIP "
                            "stream) are in flight, so kernels of different batches share the GPU" % slots) if group == 1 else
                           ("one forward of one %d-graph batch; %d independent batches (own tensors each) are served per "
                            "launch sequence - concatenated on the device by one kernel, then the eight fused kernels on "
                            "the union - and %d such groups are in flight on their own streams; K steps = K // %d group "
                            "launches + K %% %d single forwards" % (n_graphs, group, slots, group, group)),
                   "sharding": "by graph, 1 all-gather of predictions per step" if world > 1 else "single GPU"},
        "single_forward_latency_ms": latency_ms,
        "stream_placement": fwd.placement,   # in-flight streams: best of the draws is used; a random draw gives the median
        "roofline": roof,
        "forward_model": {"hbm_frac": fwd_bytes / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9),
                          "mfma_frac": fwd_flops / (ms_per_step * 1e-3) / (FP32_MFMA_PEAK_TF * 1e12),
                          "algorithmic_bytes": fwd_bytes, "flops": fwd_flops,
                          "kernels_per_forward": fwd.num_launches},
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(batch, params, DEPTH)
    if fwd.model.fused is not None:
        fwd.model.fused.release()
    args._placed_streams = list(fwd._streams) if slots > 1 else None   # the fresh-batch leg serves its calls on the same streams
    return line


# ---------------------------------------------------------------------------------------------------- config 4
def build_config4_shard(total_graphs, rank, world, seed=3456):
    """Rank ``rank``'s shard of the 100 000-molecule workload as resident ragged inputs + the shard bounds of all ranks.
    Node data come from the seeded host generator; edge counts (for the balance) and the shard's edge lists are made on
    the GPU by the engine's SetRange with config 2's rule (max_distance 4, max_neighbours 30)."""
    import torch
    from gcnn_keras_amd import sharding, synth
    from gcnn_keras_amd.graph.preprocessor import SetRange
    from gcnn_keras_amd.ragged import RaggedTensor
    nodes = synth.qm9_like_nodes(total_graphs, seed=seed)
    rule = SetRange(max_distance=4.0, max_neighbours=30)
    all_xyz = RaggedTensor.from_numpy(nodes["node_coordinates"], nodes["node_splits"])
    edge_splits = rule.count_edges(all_xyz).cpu().numpy()
    del all_xyz
    bounds = sharding.shard_bounds_by_edges(edge_splits, world)
    lo, hi = bounds[rank]
    ns = nodes["node_splits"]
    n0, n1 = int(ns[lo]), int(ns[hi])
    splits = (ns[lo:hi + 1] - n0).astype(np.int64)
    z = RaggedTensor.from_numpy(nodes["node_number"][n0:n1], splits)
    xyz = RaggedTensor.from_numpy(nodes["node_coordinates"][n0:n1], splits)
    xyz.row_splits = z.row_splits  # one partition tensor for both node properties, as a RaggedTensor batch has
    idx, _ = rule(xyz)   # carries a ready index plan: receiver-sorted, range-checked by construction
    torch.cuda.synchronize()
    return [z, xyz, idx], bounds, int(edge_splits[-1])


def run_config4(args, d, standalone=True):
    torch = d.torch
    from gcnn_keras_amd import sharding, synth
    from gcnn_keras_amd.literature import Schnet
    world = d.world
    inputs, bounds, total_edges = build_config4_shard(args.total_graphs, d.rank, world)
    n_nodes, n_edges, n_graphs = int(inputs[0].values.shape[0]), int(inputs[2].values.shape[0]), inputs[0].nrows()
    model = Schnet.make_model(depth=DEPTH)
    if d.rank == 0:
        model.set_weights(list(synth.schnet_params(seed=7).values()))
    if world > 1:            # weights replicated by one broadcast at init (rank 0 holds the seeded set)
        sharding.broadcast_weights(model, src=0)
    if model.fused is None or not model.fused.accepts(inputs):
        raise SystemExit("config 4 expects the fused route")
    gloo = world > 1 and d.backend == "gloo"

    def step(i):
        pred = model(inputs)                               # one forward of this rank's shard, no exchange inside
        if world > 1:
            pred = sharding.all_gather_predictions(pred.cpu() if gloo else pred, bounds)   # (total_graphs, 1)
        return pred

    steps, warmup = args.steps, args.warmup
    if not standalone:
        steps, warmup = 10, 3
    # binding the shard and capturing the slot's graphs (first call: direct launches; the next three: one captured graph
    # per result-ring buffer) happen here, outside the W warm-up and K timed steps, whatever W is
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    elapsed = _time_steps(d, step, warmup, steps)
    model.fused.check_flags()
    full = step(0)
    torch.cuda.synchronize()
    assert tuple(full.shape) == (args.total_graphs, 1) and bool(torch.isfinite(full).all())
    slot = model.fused.slot_of(inputs)
    roof = slot.roofline(HBM_PEAK_GBS, FP32_MFMA_PEAK_TF, iters=10)
    roof["measured"] = "HIP events on the kernel's stream around back-to-back launches of the kernel alone, after the timed region"
    roof.update(pmc_matrix_pipe(roof.get("kernel", ""), n_graphs, roof.get("avg_launch_us")))
    model.fused.release()
    if d.rank != 0:
        return None
    ms_per_step = elapsed / steps * 1e3
    fwd_flops = schnet_flops(n_nodes, n_edges, n_graphs, d=DEPTH)
    fwd_bytes = schnet_algorithmic_bytes(n_nodes, n_edges, n_graphs, d=DEPTH)
    line = {
        "metric": "edges/sec (SchNet fwd, QM9-shape batch)", "value": total_edges * steps / elapsed,
        "unit": "edges/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 4: Schnet.make_model forward (F=128, depth 3, Gauss 20) on %d QM9-shaped "
                               "molecules (seed 3456, %d directed edges) sharded by graph over %d GPU(s), balanced by edge "
                               "count; one forward per shard and step, one all-gather of the (%d,1) predictions"
                               % (args.total_graphs, total_edges, world, args.total_graphs),
                   "total_graphs": args.total_graphs, "total_edges": total_edges,
                   "rank0_shard": {"graphs": n_graphs, "nodes": n_nodes, "edges": n_edges},
                   "api": "gcnn_keras_amd.literature.Schnet.make_model(depth=3)(inputs)",
                   "sharding": ("contiguous graph ranges by edge count, no exchange during the forward, 1 "
                                "all_gather_into_tensor per step (RCCL)" if world > 1 else "single GPU, no collective")},
        "roofline": roof,
        "forward_model": {"hbm_frac": fwd_bytes / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9),
                          "mfma_frac": fwd_flops / (ms_per_step * 1e-3) / (FP32_MFMA_PEAK_TF * 1e12),
                          "algorithmic_bytes_rank0": fwd_bytes, "flops_rank0": fwd_flops},
    }
    if standalone and world == 1 and not args.no_cpu_baseline:
        sample = synth.qm9_like_batch(num_graphs=128, seed=3456)   # the first 128 molecules of the same stream
        line["cpu_baseline"] = cpu_baseline(sample, params, DEPTH)
    return line


# ---------------------------------------------------------------------------------------------------- fresh batches
def _graph_list(batch):
    """A synth batch as the reference's list of per-graph dicts (what ``MemoryGraphList`` holds, kgcnn/data/base.py)."""
    ns, es = batch["node_splits"], batch["edge_splits"]
    return [{"node_number": batch["node_number"][ns[g]:ns[g + 1]], "node_coordinates": batch["node_coordinates"][ns[g]:ns[g + 1]],
             "edge_indices": batch["edge_indices"][es[g]:es[g + 1]]} for g in range(len(ns) - 1)]


def run_stream(args, d, batches=64, in_flight=4):
    """The reference's use of a model is ``model.predict`` over DIFFERENT batches (kgcnn/data/base.py:203-239): every batch
    is seen once.  ``batches`` distinct 128-graph batches (different N, M; seeds 1234 + k), each called ONCE through
    ``Schnet.make_model(...)(inputs)`` - first sight = bind (buffers from the route's arena, no index pass: the packer
    classified the list) + one direct launch of the eight kernels, no graph capture, no replay - ``in_flight`` calls under
    as many streams.  Two timings: (a) batches already resident in HBM (the contract of ``value``), (b) including the host
    packer (``BatchPacker``: concatenate into pinned staging, index plan, async H2D), which overlaps with the GPU work."""
    torch = d.torch
    from gcnn_keras_amd import synth
    from gcnn_keras_amd.data.packer import BatchPacker
    from gcnn_keras_amd.literature import Schnet
    items = [{"name": "node_number", "ragged": True, "dtype": "float32"},
             {"name": "node_coordinates", "ragged": True, "dtype": "float32"},
             {"name": "edge_indices", "ragged": True, "dtype": "int64"}]
    lists, edges = [], []
    for k in range(batches):
        b = synth.qm9_like_batch(num_graphs=args.graphs, seed=1234 + k)
        lists.append(_graph_list(b))
        edges.append(int(b["edge_splits"][-1]))
    model = Schnet.make_model(depth=DEPTH)
    model.set_weights(list(synth.schnet_params(seed=7).values()))
    if in_flight > 1 and os.environ.get("MPENGINE_INFLIGHT_NODE_HALF", "1") != "0":
        model.fused.cfconv_flags |= 512     # several launch sequences in flight (as engine.SchnetForward sets it)
    placed = getattr(args, "_placed_streams", None)
    streams = placed[:in_flight] if placed and len(placed) >= in_flight else [torch.cuda.Stream() for _ in range(in_flight)]
    packer = BatchPacker(items, index_item="edge_indices", node_item="node_number", slots=2 * in_flight)

    def as_inputs(pb):
        return [pb["node_number"], pb["node_coordinates"], pb["edge_indices"]]

    # warm-up: module load, weight images, allocator pools of every stream (batches that are NOT part of the timed sets)
    warm = synth.qm9_like_batch(num_graphs=args.graphs, seed=99)
    for s in streams:
        for _ in range(3):
            pb = packer.pack(_graph_list(warm))
            with torch.cuda.stream(s):
                pb.wait(s)
                model(as_inputs(pb))
    torch.cuda.synchronize()
    out = {"batches": batches, "graphs_per_batch": args.graphs, "in_flight": in_flight, "edges": int(sum(edges))}
    # (b) with the host packer in the loop (run first: freeing the 64 resident batches of the other legs - device blocks that
    #     were used on several streams - leaves the caching allocator with event bookkeeping that slows the next allocations)
    results = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, g in enumerate(lists):
        pb = packer.pack(g)
        s = streams[k % in_flight]
        with torch.cuda.stream(s):
            pb.wait(s)
            results.append(model(as_inputs(pb)))
    torch.cuda.synchronize()
    t_pack = time.perf_counter() - t0
    out.update({"edges_per_s_with_host_packing": sum(edges) / t_pack, "ms_per_batch_with_host_packing": t_pack / batches * 1e3})
    model.fused.release()
    del results
    # (a) resident: everything packed and copied before the clock starts
    resident = [packer.pack(g) for g in lists[:2 * in_flight]]   # the packer's staging slots bound how many may be alive
    del resident
    packer_all = BatchPacker(items, index_item="edge_indices", node_item="node_number", slots=batches)
    resident = [packer_all.pack(g) for g in lists]
    for pb in resident:
        pb.wait(torch.cuda.current_stream())
    torch.cuda.synchronize()
    host_us, results = [], []
    t0 = time.perf_counter()
    ins = [as_inputs(pb) for pb in resident]
    base = torch.cuda.current_stream()
    t0 = time.perf_counter()
    for k, x in enumerate(ins):
        torch.cuda.set_stream(streams[k % in_flight])     # (a stream context per call costs ~10 us of host time)
        h0 = time.perf_counter()
        results.append(model(x))
        host_us.append((time.perf_counter() - h0) * 1e6)
    torch.cuda.set_stream(base)
    torch.cuda.synchronize()
    t_res = time.perf_counter() - t0
    assert model.fused is not None and model.fused.last == "direct"
    model.fused.check_flags()
    assert all(bool(torch.isfinite(r).all()) for r in results)
    out.update({"edges_per_s_resident": sum(edges) / t_res, "ms_per_batch_resident": t_res / batches * 1e3,
                "host_us_per_first_call_median": float(np.median(host_us)),
                "host_us_per_first_call_p90": float(np.percentile(host_us, 90))})
    # (a') the same resident batches, served five per launch sequence (route.call_group: one concatenation launch + the
    #      eight kernels for five never-seen batches instead of forty launches)
    model.fused.release()
    grp = 5
    for w in range(2):                                    # arena / allocator pools of the group-sized buffers
        for k in range(0, 2 * grp * in_flight, grp):
            torch.cuda.set_stream(streams[(k // grp) % in_flight])
            model.fused.call_group(ins[k:k + grp])
        torch.cuda.set_stream(base)
        torch.cuda.synchronize()
        model.fused.release() if w == 0 else None
    model.fused._groups.clear()
    results = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(0, batches - batches % grp, grp):
        torch.cuda.set_stream(streams[(k // grp) % in_flight])
        results.extend(model.fused.call_group(ins[k:k + grp]))
    for k in range(batches - batches % grp, batches):
        results.append(model(ins[k]))
    torch.cuda.set_stream(base)
    torch.cuda.synchronize()
    t_grp = time.perf_counter() - t0
    assert len(results) == batches and all(bool(torch.isfinite(r).all()) for r in results)
    out.update({"edges_per_s_resident_grouped": sum(edges) / t_grp, "ms_per_batch_resident_grouped": t_grp / batches * 1e3,
                "batches_per_launch_group": grp})
    del resident, results, packer_all, ins
    model.fused.release()
    model.fused.release()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="auto", choices=["auto", "config2", "config4", "stream"],
                    help="auto: config2 at 1 GPU (the configuration the metric is quoted on), config4 (100 000 molecules "
                         "sharded by graph + one all-gather) at N > 1")
    ap.add_argument("--graphs", type=int, default=128, help="config2: graphs per GPU (BASELINE config 2 = 128)")
    ap.add_argument("--total-graphs", type=int, default=100000, help="config4: molecules in the whole job")
    ap.add_argument("--mode", default="auto", choices=["auto", "fused", "layers"])
    ap.add_argument("--in-flight", type=int, default=None,
                    help="config2: independent batches - or launch groups - (own tensors + batch slot + HIP stream) whose "
                         "forwards overlap on the GPU; 1 = strictly one at a time; default 4")
    ap.add_argument("--group", type=int, default=None,
                    help="config2: independent batches served by ONE launch sequence (concatenated on the device, "
                         "route.call_group); 1 = every batch its own model(inputs) call; default 5 (1 when --in-flight is given alone)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream", action="store_true",
                    help="skip the fresh-batch measurement (64 distinct batches, each called once) appended to the default line")
    ap.add_argument("--no-config4-reference", action="store_true",
                    help="skip the single-GPU config-4 rate appended to the default 1-GPU line")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production); gloo only rehearses N > 1 on a single-GPU box")
    args = ap.parse_args()

    # Several forwards are kept in flight on separate HIP streams (--in-flight).  ROCm multiplexes streams onto a few
    # hardware queues; 8 instead of the default 4 leave room beside the null stream, and SchnetForward.load_batch picks
    # the best of a few stream draws (engine.py:_place_streams).  Must be set before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    d = _Dist(args)
    workload = args.workload
    if workload == "auto":
        workload = "config2" if d.world == 1 else "config4"
    if workload == "stream":
        if d.world != 1:
            raise SystemExit("--workload stream is a single-GPU measurement")
        print(json.dumps({"stream_fresh_batches": run_stream(args, d)}))
        d.close()
        return
    if workload == "config2":
        line = run_config2(args, d)
        if line is not None and d.world == 1 and args.workload == "auto" and not args.no_stream:
            fresh = run_stream(args, d)
            fresh["fraction_of_replay_value"] = fresh["edges_per_s_resident"] / line["value"]
            fresh["grouped_fraction_of_replay_value"] = fresh["edges_per_s_resident_grouped"] / line["value"]
            line["stream_fresh_batches"] = fresh
        if line is not None and d.world == 1 and args.workload == "auto" and not args.no_config4_reference:
            ref = run_config4(args, d, standalone=False)
            line["config4_single_gpu"] = {k: ref[k] for k in ("value", "unit", "ms_per_step", "steps", "scaling")}
            line["config4_single_gpu"]["workload"] = ref["config"]["workload"]
            line["config4_single_gpu"]["cfconv_tflops"] = ref["roofline"]["achieved"]
    else:
        line = run_config4(args, d)
    if d.rank == 0:
        print(json.dumps(line))
    d.close()


if __name__ == "__main__":
    main()
