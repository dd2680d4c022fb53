"""Merge fresh --pmc passes (scripts/run_pmc_gcn.sh: gpurun_out/pmc3_gcn_{fetch,write}, pmc2_{fetch,write}) into
profiles/r03_pmc_hbm_traffic.json (round 3: + the launch-group run, scripts/run_pmc_groups.sh): entries with the same
label are replaced, the others kept.

    python scripts/merge_pmc.py
"""
import collections
import csv
import glob
import json
import os
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "r03_pmc_hbm_traffic.json")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def agg(d):
    files = (glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv"))
             + glob.glob(os.path.join(ROOT, "gpurun_out", d, "*counter_collection.csv")))
    acc = collections.defaultdict(list)
    if files:
        rows = list(csv.DictReader(open(files[0])))
        biggest = collections.defaultdict(int)
        for r in rows:
            biggest[short(r["Kernel_Name"])] = max(biggest[short(r["Kernel_Name"])], int(r["Grid_Size"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            # a launch-group run also issues single-batch launches of the same kernels: only the union launches count
            if "_grp_" in d and int(r["Grid_Size"]) != biggest[k]:
                continue
            acc[k].append(float(r["Counter_Value"]))
    return acc


def entry(fetch_dir, write_dir, label):
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


new = [entry("pmc2_fetch", "pmc2_write", "config 2 (128 graphs, N=2301, M=26190): bench.py --in-flight 1, forward"),
       entry("pmc_grp_fetch", "pmc_grp_write", "config 2 groups (five 128-graph batches per launch sequence: 640 graphs, "
             "N=11.5 k, M=131 k per launch): bench.py --in-flight 1 --group 5"),
       entry("pmc3_gcn_fetch", "pmc3_gcn_write", "config 5 (Cora-shaped graph, N=2708, M=13264, F=1433): "
             "scripts/profile_gcn.py, fused GCN forward")]
new = [e for e in new if e["kernels"]]
old = json.load(open(OUT)) if os.path.exists(OUT) else []
labels = {e["label"] for e in new}
merged = [e for e in old if e["label"] not in labels] + new
merged.sort(key=lambda e: e["label"])
json.dump(merged, open(OUT, "w"), indent=1)
for e in new:
    print(e["label"])
    for k, v in e["kernels"].items():
        print("   %-62s fetch %9.1f KB (x2 = %9.1f)  write %9.1f KB" % (k, v["FETCH_SIZE_KB_median"],
                                                                         2 * v["FETCH_SIZE_KB_median"],
                                                                         v["WRITE_SIZE_KB_median"]))
