#!/usr/bin/env python3
"""Turns the scratch output of tools/profile.sh (gpurun_out/prof), tools/pmc_collect.sh (gpurun_out/pmc_summary.csv) and the side
measurement tools into the committed, judged summaries under profiles/ for a round: python tools/make_profiles.py r03"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

for name, what in (("trace", "python3 bench.py --no-cpu --no-extra  (only the headline launches)"), ("trace_full", "python3 bench.py  (all legs)")):
    rows = list(csv.DictReader(open(os.path.join(src, f"kernel_stats_{name}.csv"))))
    suffix = "" if name == "trace" else "_full"
    with open(os.path.join(dst, f"{tag}_kernel_stats{suffix}.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {what}   (MI355X, 1 GPU; kernel names cut at 110 chars)\n")
        w = csv.writer(f)
        keys = list(rows[0].keys())
        w.writerow(keys)
        for r in rows:
            w.writerow([r[k][:110] if k == "Name" else r[k] for k in keys])

raw = json.load(open(os.path.join(src, "pmc_traffic_raw.json")))
fetch_kb, write_kb = raw["FETCH_SIZE"]["mean_per_launch"], raw["WRITE_SIZE"]["mean_per_launch"]
bench = json.loads(open(os.path.join(src, "bench_under_trace.json")).read().strip().splitlines()[-1])
full = json.loads(open(os.path.join(src, "bench_full_under_trace.json")).read().strip().splitlines()[-1])
traffic = {
    "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu --no-extra --steps 5 --warmup 2",
    "kernel": "gms::filter_kernel_dense<10, false, 1024, false>",
    "FETCH_SIZE_raw_KB_per_launch": fetch_kb,
    "WRITE_SIZE_raw_KB_per_launch": write_kb,
    "gfx950_correction": "FETCH_SIZE counts 64 B per 128-B request: x2 (MI355X_MICROARCH.md, HBM section); confirmed for this kernel's "
                         "read shape (16 B/lane) by tools/ubench/fetch_calib.hip on a 2 GiB buffer in round 1: ratio 0.50000. "
                         "WRITE_SIZE is exact for 16-B-per-lane stores.",
    "hbm_read_bytes_per_launch": fetch_kb * 1024 * 2,
    "hbm_write_bytes_per_launch": write_kb * 1024,
    "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "pairs_per_launch": bench["config"]["pairs_per_step_per_gpu"],
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
for obj in (bench, full):
    obj["roofline"].pop("traffic_from_profiles", None)  # (the previous round's file, as the run found it)
with open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w") as f:
    f.write(json.dumps(bench) + "\n" + json.dumps(full) + "\n")
sq = os.path.join(ROOT, "gpurun_out", "pmc_summary.csv")
if os.path.exists(sq):
    shutil.copy(sq, os.path.join(dst, f"{tag}_sq_counters.csv"))
print(json.dumps({k: traffic[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch")}))
