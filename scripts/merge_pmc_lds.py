"""Condense the LDS counter pass of scripts/run_pmc_nodes_r03.sh (gpurun_out/pmc_lds_grp) into profiles/r03_pmc_lds.json:
per kernel the median per-launch value of SQ_INSTS_LDS, SQ_LDS_IDX_ACTIVE (all LDS-array cycles), SQ_LDS_BANK_CONFLICT
(extra cycles through bank conflicts), SQ_LDS_ADDR_CONFLICT, SQ_LDS_UNALIGNED_STALL - for the union launches of a launch
group (the launches with a kernel's largest grid)."""
import collections
import csv
import glob
import json
import os
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]


def main():
    files = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_lds_grp", "*counter_collection.csv"))
    if not files:
        print("no LDS pass under gpurun_out/")
        return
    rows = list(csv.DictReader(open(files[0])))
    big = collections.defaultdict(int)
    for r in rows:
        big[short(r["Kernel_Name"])] = max(big[short(r["Kernel_Name"])], int(r["Grid_Size"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = short(r["Kernel_Name"])
        if int(r["Grid_Size"]) == big[k] and not any(s in k for s in ("rocclr", "at::", "pack")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"label": "config 2 groups (five 128-graph batches per launch sequence: 640 graphs, N=11.5 k, M=131 k per launch): "
                    "bench.py --in-flight 1 --group 5, union launches",
           "note": "one --pmc pass; medians per launch.  SQ_LDS_BANK_CONFLICT = extra LDS cycles through bank conflicts "
                   "(MI355X_MICROARCH.md, LDS): 0 for the node chains' bf16-plane tiles (DESIGN 3.6)",
           "kernels": {k: dict({"launches": max(len(x) for x in v.values())},
                               **{n: statistics.median(x) for n, x in sorted(v.items())}) for k, v in sorted(acc.items())}}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_pmc_lds.json"), "w"), indent=1)
    for k, v in out["kernels"].items():
        print(k, v)


if __name__ == "__main__":
    main()
