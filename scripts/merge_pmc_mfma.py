"""Condense the --pmc passes of scripts/run_pmc_mfma.sh (gpurun_out/pmc_mfma_*) into profiles/r03_pmc_mfma.json:
per kernel the median per-launch value of every collected counter and the derived figures

  kernel_cycles     = GRBM_GUI_ACTIVE / 8          (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md, DVFS note)
  matrix_pipe_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel_cycles)
  valu_per_mfma_cyc = SQ_INSTS_VALU / SQ_VALU_MFMA_BUSY_CYCLES

SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles per instruction (32 per v_mfma_f32_32x32x16_bf16, the guide's
constants table), summed over all SIMDs.  bench.py reads ``matrix_pipe_busy`` of the dominant kernel from this file.
"""
import collections
import csv
import glob
import json
import os
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "r03_pmc_mfma.json")
PASSES = [("pmc_mfma_c2", "config 2 (128 graphs, N=2301, M=26190): bench.py --in-flight 1, forward"),
          ("pmc_mfma_grp", "config 2 groups (five 128-graph batches per launch sequence: 640 graphs, N=11.5 k, M=131 k per "
                           "launch): bench.py --in-flight 1 --group 5"),
          ("pmc_mfma_shard", "12500 graphs (config-4 shard, N=225 k, M=2.56 M): bench.py --workload config4 --total-graphs 12500"),
          ("pmc_mfma_painn", "config 3 (PaiNN, 64 graphs, N=1344, M=20586): scripts/profile_painn.py force")]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def collect(d):
    """kernel -> counter -> values.  A launch-group run (directory name ending in "_grp") also issues single-batch launches
    of the same kernels (latency loop, warm-up): only the launches with a kernel's LARGEST grid - the union launches - count."""
    files = (glob.glob(os.path.join(ROOT, "gpurun_out", d, "*", "*counter_collection.csv"))
             + glob.glob(os.path.join(ROOT, "gpurun_out", d, "*counter_collection.csv")))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files[:1]:
        rows = list(csv.DictReader(open(f)))
        biggest = collections.defaultdict(int)
        for r in rows:
            biggest[short(r["Kernel_Name"])] = max(biggest[short(r["Kernel_Name"])], int(r["Grid_Size"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            if d.endswith("_grp") and int(r["Grid_Size"]) != biggest[k]:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    entries = []
    for d, label in PASSES:
        acc = collect(d)
        kernels = {}
        for k in sorted(acc):
            if "rocclr" in k or "at::" in k or "pack" in k or "rocprim" in k:
                continue
            c = {n: statistics.median(v) for n, v in acc[k].items()}
            row = {"launches": max(len(v) for v in acc[k].values())}
            row.update({n: c[n] for n in sorted(c)})
            gui, busy = c.get("GRBM_GUI_ACTIVE"), c.get("SQ_VALU_MFMA_BUSY_CYCLES")
            if gui:
                row["kernel_cycles"] = gui / 8.0
                if busy is not None:
                    row["matrix_pipe_busy"] = busy / (1024.0 * gui / 8.0)
            if busy and c.get("SQ_INSTS_VALU") is not None:
                row["valu_insts_per_mfma_busy_cycle"] = c["SQ_INSTS_VALU"] / busy
            kernels[k] = row
        if kernels:
            entries.append({"label": label, "kernels": kernels,
                            "note": "one --pmc pass (SQ block: 7 counters, GRBM: 1); medians per launch; SQ_WAVE_CYCLES / "
                                    "SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles "
                                    "(MI355X_MICROARCH.md constants table)"})
    if not entries:
        print("no pmc_mfma passes under gpurun_out/")
        return
    old = json.load(open(OUT)) if os.path.exists(OUT) else []
    labels = {e["label"] for e in entries}
    merged = [e for e in old if e["label"] not in labels] + entries
    json.dump(merged, open(OUT, "w"), indent=1)
    for e in entries:
        print(e["label"])
        for k, v in e["kernels"].items():
            print("   %-60s busy %.3f  cycles %9.0f  mfma_busy %12.0f  valu %10.0f" % (
                k, v.get("matrix_pipe_busy", float("nan")), v.get("kernel_cycles", float("nan")),
                v.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")), v.get("SQ_INSTS_VALU", float("nan"))))


if __name__ == "__main__":
    main()
