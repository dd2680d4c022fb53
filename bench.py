#!/usr/bin/env python
"""bench.py - edges/sec of a SchNet forward on QM9-shaped synthetic batches (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one SchNet forward (kgcnn.literature.Schnet.make_model, F=128, depth 3, Gauss(20, 4.0, 0.4), sum pooling)
over one resident batch of BASELINE config 2 (128 QM9-shaped graphs, seed 1234 + rank): raw API inputs (float node
numbers, float32 coordinates, int64 (M,2) sample indices, int64 row_splits) are in HBM when the clock starts; the index
preparation (shift, CSR), geometry, basis expansion, every interaction block, readout and output MLP are inside it.
Graphs shard by rank with no data-path collective (weak scaling: every rank owns its own 128-graph batch); the only
collective is one all-gather of the (G,1) predictions per step (RCCL over xGMI), as BASELINE.json's north_star states.

One 128-graph forward cannot fill an MI355X (819 edge tiles for 1024 SIMDs, 144 node tiles for 256 CUs), so the loop keeps
`--in-flight` (default 4) independent batch slots - own buffers, own HIP stream, own captured graph - busy: step i is one
full forward of slot i % 4, launched with one hipGraphLaunch; kernels of different slots overlap on the GPU.  Exactly K
forwards run inside the timed region.  `single_forward_latency_ms` reports the latency of a lone forward beside the
throughput; `--in-flight 1` runs strictly one forward at a time.

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` (dominant kernel, measured live with HIP events)
and `cpu_baseline` (the NumPy oracle restating the reference's unfused TF op sequence, timed on the host cores).
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


def schnet_algorithmic_bytes(n, m, g, f=128, b=20, d=3):
    """bytes = [16N + 16M + 4MB + 4NF] + D[8NF + 4MB + 16M] + [4NF + 4G]."""
    return (16 * n + 16 * m + 4 * m * b + 4 * n * f) + d * (8 * n * f + 4 * m * b + 16 * m) + (4 * n * f + 4 * g)


def schnet_flops(n, m, g, f=128, b=20, d=3):
    """flops = M D [2(BF + F^2) + 2F] + N [2*64F + D 6F^2 + 2(F^2 + 64F)] + G 2(64^2 + 64)."""
    return (m * d * (2 * (b * f + f * f) + 2 * f) + n * (2 * 64 * f + d * 6 * f * f + 2 * (f * f + 64 * f))
            + g * 2 * (64 * 64 + 64))


def cpu_baseline(batch, params, depth, budget_s=12.0):
    """The CPU port of the reference's unfused op sequence, timed on this host's cores; bounded to ~budget_s.
    Prefers the C/OpenMP port (oracle/mp_oracle.c, all host cores); falls back to the NumPy oracle."""
    from oracle import c_oracle
    from oracle import kgcnn_oracle as ko
    m = int(batch["edge_splits"][-1])
    g = len(batch["node_splits"]) - 1
    if c_oracle.available():
        def run():
            return c_oracle.schnet_forward(params, batch["node_number"], batch["node_coordinates"],
                                           batch["edge_indices"], batch["node_splits"], batch["edge_splits"],
                                           depth=depth)
        cores, what = c_oracle.num_threads(), "C/OpenMP port oracle/mp_oracle.c"
    else:
        inputs = (ko.R(batch["node_number"], batch["node_splits"]),
                  ko.R(batch["node_coordinates"], batch["node_splits"]),
                  ko.R(batch["edge_indices"], batch["edge_splits"]))

        def run():
            return ko.schnet_forward(params, *inputs, depth=depth)
        try:
            import threadpoolctl
            cores = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
        except Exception:  # pragma: no cover
            cores = os.cpu_count() or 1
        what = "NumPy oracle (BLAS sgemm threaded, gather / segment ops single-threaded)"
    run()  # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < 200:
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": m / med, "unit": "edges/s", "cores": int(cores), "kind": "port",
            "sample": "%d forwards of the same %d-graph batch (median %.1f ms) with the %s on %d threads"
                      % (len(times), g, med * 1e3, what, int(cores))}


def pmc_traffic(kernel_name, graphs):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*pmc*.json):
    FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command; FETCH doubled as the microarch guide
    prescribes for gfx950.  None when no pass exists for this workload size."""
    import glob
    want = {128: "config 2", 12500: "12500 graphs"}.get(graphs)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic.json")), reverse=True):
        try:
            for entry in json.load(open(path)):
                if want and entry["label"].startswith(want):
                    for k, v in entry["kernels"].items():
                        if k.split("<")[0] in kernel_name:
                            raw_f, raw_w = v["FETCH_SIZE_KB_median"] * 1024, v["WRITE_SIZE_KB_median"] * 1024
                            return {"traffic": 2 * raw_f + raw_w,
                                    "traffic_detail": {"source": os.path.basename(path), "fetch_bytes_raw": raw_f,
                                                       "write_bytes": raw_w, "fetch_correction": "x2 (gfx950)"}}
        except Exception:
            continue
    return {"traffic": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graphs", type=int, default=128, help="graphs per GPU (BASELINE config 2 = 128)")
    ap.add_argument("--mode", default="auto", choices=["auto", "fused", "layers"])
    ap.add_argument("--in-flight", type=int, default=4,
                    help="independent batch slots (own buffers + HIP stream + graph) whose forwards overlap on the GPU; "
                         "1 = strictly one forward at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production); gloo only rehearses N > 1 on a single-GPU box")
    args = ap.parse_args()

    # Several forwards are kept in flight on separate HIP streams (--in-flight).  ROCm multiplexes streams onto a few
    # hardware queues; 8 instead of the default 4 leave room beside the null stream, and SchnetForward.load_batch picks
    # the best of a few stream draws (engine.py:_place_streams).  Must be set before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    from gcnn_keras_amd import synth
    from gcnn_keras_amd.engine import SchnetForward

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    ndev = torch.cuda.device_count()
    device_index = local_rank % max(ndev, 1)  # one rank per GPU on a full node; wraps only in single-GPU rehearsals
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    depth = 3
    batch = synth.qm9_like_batch(num_graphs=args.graphs, seed=1234 + rank)
    params = synth.schnet_params(seed=7)  # Keras defaults: glorot_uniform kernels, zero biases, U(-0.05, 0.05) embedding
    n_nodes, n_edges, n_graphs = int(batch["node_splits"][-1]), int(batch["edge_splits"][-1]), args.graphs

    fwd = SchnetForward(params, depth=depth, mode=args.mode, in_flight=args.in_flight)
    fwd.load_batch(batch)
    slots = fwd.in_flight
    gathered = [torch.empty((world * n_graphs, 1), dtype=torch.float32, device="cuda") for _ in range(slots)] \
        if world > 1 else None

    host_parts = [torch.empty((n_graphs, 1)) for _ in range(world)] if (world > 1 and args.backend == "gloo") else None

    def step(i):
        # one forward of one batch on its slot's stream; with RCCL the all-gather is ordered after it on that stream
        # (the process group serialises the collectives of different slots in issue order)
        if world == 1:
            return fwd.replay(i)                                       # one C-ABI call: hipGraphLaunch on the slot's stream
        torch.cuda.set_stream(fwd.stream_of(i))                        # cheaper than a stream context per step
        out = fwd.replay(i)
        if host_parts is None:
            dist.all_gather_into_tensor(gathered[i % slots], out)
        else:
            dist.all_gather(host_parts, out.cpu())                     # rehearsal only
        return out

    def reduce_scalar(value, op):
        t = torch.tensor([value], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=op)
        return float(t.item())

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.set_stream(torch.cuda.default_stream())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = reduce_scalar(elapsed, dist.ReduceOp.MAX if world > 1 else None)
    total_edges = reduce_scalar(float(n_edges), dist.ReduceOp.SUM if world > 1 else None)

    fwd.check_flags()
    # latency of ONE forward with nothing else on the GPU (slot 0 alone), for reference beside the throughput
    with torch.cuda.stream(fwd.stream_of(0)):
        for _ in range(10):
            fwd.replay(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(100):
            fwd.replay(0)
        torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) / 100 * 1e3
    roof = fwd.roofline(HBM_PEAK_GBS, FP32_MFMA_PEAK_TF)  # dominant kernel, timed with HIP events on its stream
    roof.update(pmc_traffic(roof.get("kernel", ""), n_graphs))
    roof["measured"] = ("HIP events on the kernel's stream around back-to-back launches of the kernel alone on the GPU, after "
                        "the timed region; rocprofv3 of `bench.py --in-flight 1` (profiles/r01_fused_config2_kernel_stats.csv) "
                        "gives the same average; with batches in flight kernels of different batches share CUs and their "
                        "individual durations stretch (profiles/r01_fused_config2_inflight4_kernel_stats.csv)")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_edges * args.steps / elapsed
        fwd_bytes = schnet_algorithmic_bytes(n_nodes, n_edges, n_graphs, d=depth)
        fwd_flops = schnet_flops(n_nodes, n_edges, n_graphs, d=depth)
        line = {
            "metric": "edges/sec (SchNet fwd, QM9-shape batch)", "value": value, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: SchNet forward (F=128, depth 3, Gauss 20) on %d QM9-shaped graphs "
                                   "per GPU, N=%d nodes, M=%d directed edges on rank 0" % (n_graphs, n_nodes, n_edges),
                       "graphs_per_gpu": n_graphs, "nodes": n_nodes, "edges": n_edges, "mode": fwd.mode,
                       "in_flight": slots,
                       "step": "one forward of one batch; %d independent batch slots (own buffers, HIP stream and "
                               "graph) are in flight, so kernels of different batches share the GPU" % slots,
                       "sharding": "by graph, 1 all-gather of predictions per step" if world > 1 else "single GPU"},
            "single_forward_latency_ms": latency_ms,
            "roofline": roof,
            "forward_model": {"hbm_frac": fwd_bytes / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9),
                              "mfma_frac": fwd_flops / (ms_per_step * 1e-3) / (FP32_MFMA_PEAK_TF * 1e12),
                              "algorithmic_bytes": fwd_bytes, "flops": fwd_flops, "kernels_per_forward": fwd.num_launches},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batch, params, depth)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
