#!/bin/bash
# the driver's command three times and the default run once; one summary line each
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_driver_$i.log 2>&1; done
python bench.py > gpurun_out/final_default.log 2>&1
python - <<PY
import json
for f in ["final_driver_1", "final_driver_2", "final_driver_3", "final_default"]:
    d = json.loads([l for l in open("gpurun_out/%s.log" % f) if l.startswith("{")][-1])
    s, r = d["stream_fresh_batches"], d["roofline"]
    print(f, round(d["value"] / 1e6), round(d["ms_per_step"] * 1e3, 2), "lone", round(d["single_forward_latency_ms"] * 1e3, 1),
          "roof", r["kernel"][:28], round(r["frac"], 3), round(r["avg_launch_us"], 1), "traffic", r["traffic"], "pipe",
          round(r.get("matrix_pipe_pmc", {}).get("matrix_pipe_busy_at_2p4GHz", 0), 3), "fresh",
          round(s["edges_per_s_resident"] / 1e6), round(s["edges_per_s_resident_grouped"] / 1e6),
          round(s["grouped_fraction_of_replay_value"], 2), "c4", round(d["config4_single_gpu"]["value"] / 1e6), "cpu",
          round(d["cpu_baseline"]["value"] / 1e6, 2), "median draw", round(d["stream_placement"]["median_us_per_step"], 2))
PY
