"""Condense rocprofv3 outputs under gpurun_out/ into the tracked summaries under profiles/ (run after a profiling call).

    python scripts/summarize_profiles.py r01
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def kernel_stats(run_dir, out_name, title):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", run_dir, "*", "*kernel_stats.csv"))
    if not files:
        return
    rows = list(csv.DictReader(open(files[0])))
    with open(os.path.join(ROOT, "profiles", out_name), "w") as f:
        f.write("# %s\n# source: rocprofv3 --kernel-trace --stats (gpurun_out/%s), MI355X\n" % (title, run_dir))
        f.write("kernel,calls,avg_us,min_us,max_us,percent\n")
        for r in rows:
            if "at::native" in r["Name"]:
                continue
            f.write("%s,%s,%.2f,%.2f,%.2f,%s\n" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                   float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))


def pmc(fetch_dir, write_dir, label):
    def agg(d):
        files = (glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv"))
                 + glob.glob(os.path.join(ROOT, "gpurun_out", d, "*counter_collection.csv")))
        acc = collections.defaultdict(list)
        if files:
            for r in csv.DictReader(open(files[0])):
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        return acc
    fa, wa = agg(fetch_dir), agg(write_dir)
    out = {}
    for k in sorted(fa):
        if "rocclr" in k or "pack" in k or "at::" in k:
            continue
        out[k] = {"launches": len(fa[k]), "FETCH_SIZE_KB_median": statistics.median(fa[k]),
                  "WRITE_SIZE_KB_median": statistics.median(wa.get(k, [0.0]))}
    return {"label": label, "kernels": out,
            "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); values in KB per launch as reported. On gfx950 "
                    "FETCH_SIZE under-counts wide (16 B/lane) coalesced reads by exactly 2x (MI355X_MICROARCH.md, HBM); "
                    "WRITE_SIZE is exact for 16-B stores and float atomics."}


os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
if tag != "r01":   # round 2 on: kernel statistics come from rocprof_db_stats.py; this script condenses the --pmc passes
    res = [pmc("pmc2_fetch", "pmc2_write", "config 2 (128 graphs, N=2301, M=26190): bench.py --in-flight 1, forward"),
           pmc("pmc2_sf_fetch", "pmc2_sf_write", "scripts/bench_schnet_force.py 64 --profile fork: SchNet energy+force, "
               "fork configuration (depth 6, 25 bins), N=1344, M=20586"),
           pmc("pmc2_pn_fetch", "pmc2_pn_write", "scripts/profile_painn.py force: PaiNN energy+force, config 3, N=1344, "
               "M=20586")]
    res.append(pmc("pmc3_gcn_fetch", "pmc3_gcn_write", "config 5 (Cora-shaped graph, N=2708, M=13264, F=1433): "
                   "scripts/profile_gcn.py, fused GCN forward"))
    # passes that were not run this time keep their committed entry (same merge rule as scripts/merge_pmc.py)
    out_path = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.json" % tag)
    res = [e for e in res if e["kernels"]]
    old = json.load(open(out_path)) if os.path.exists(out_path) else []
    fresh = {e["label"] for e in res}
    merged = sorted([e for e in old if e["label"] not in fresh] + res, key=lambda e: e["label"])
    json.dump(merged, open(out_path, "w"), indent=1)
    print("profiles written")
    sys.exit(0)
kernel_stats("prof_layers", "%s_layers_mode_kernel_stats.csv" % tag, "bench.py --mode layers (one engine call per Keras layer), config 2")
kernel_stats("prof_fused4", "%s_fused_config2_kernel_stats.csv" % tag, "bench.py --in-flight 1 (fused, HIP graph, one forward at a time), config 2: 128 graphs")
kernel_stats("prof_inflight", "%s_fused_config2_inflight4_kernel_stats.csv" % tag, "bench.py (default: 4 batches in flight on 4 streams; kernel durations include time shared with other batches), config 2")
kernel_stats("prof_big", "%s_fused_12500graphs_kernel_stats.csv" % tag, "bench.py --graphs 12500 (config-4 shard size), fused")
res = [pmc("pmc_fetch", "pmc_write", "config 2 (128 graphs, N=2301, M=26190)"),
       pmc("pmc_fetch_big", "pmc_write_big", "12500 graphs (N=225225, M=2556724)")]
json.dump(res, open(os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.json" % tag), "w"), indent=1)
print("profiles written")
