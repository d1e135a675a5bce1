#!/usr/bin/env python3
"""After tools/profile.sh, tools/pmc_collect.sh (twice) and the side-measurement tools have written gpurun_out/: copy what is to be judged
into profiles/ (through tools/make_profiles.py) and print the numbers the documents quote. python tools/collect_final.py [r02]"""
import csv, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
os.chdir(ROOT)
subprocess.check_call([sys.executable, "tools/make_profiles.py", tag])
F = "gpurun_out/final/"
for src, dst in (("bench.json", "bench.json"), ("phase.json", "phase_cycles.json"), ("phase_rs.json", "phase_cycles_rot_scale.json"),
                 ("phase_rs_noprobe.json", "phase_cycles_rot_scale_probe_off.json"), ("bench_2ranks.json", "bench_2ranks_one_device_rehearsal.json"),
                 ("bench_spatial.json", "bench_spatially_ordered_keypoints.json")):
    shutil.copy(F + src, f"profiles/{tag}_{dst}")
shutil.copy("gpurun_out/pmc_summary_ord.csv", f"profiles/{tag}_sq_counters_spatially_ordered_keypoints.csv")
for k in ("orb", "sift"):
    shutil.copy(f"gpurun_out/bf_pmc_{k}/summary.csv", f"profiles/{tag}_matcher_counters_{k}.csv")
misc = json.load(open(F + "misc.json"))
misc["crowded_bench"] = json.load(open(F + "crowded.json"))
misc["rot_scale_bench_512_per_launch"] = json.load(open(F + "rs512.json"))
misc["rot_scale_bench_8192_per_launch"] = json.load(open(F + "rs8192.json"))
misc["rot_scale_bench_512_per_launch_GMS_SCALE_PROBE_0"] = json.load(open(F + "rs512_noprobe.json"))
json.dump(misc, open(f"profiles/{tag}_side_measurements.json", "w"), indent=1)
d = json.loads(open(F + "bench.json").read().strip().splitlines()[-1])
print("HEAD", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["achieved"])
print("CPU", {k: round(v["pairs_per_s"]) for k, v in d["cpu_baseline"]["by_threads"].items()}, d["gpu_vs_cpu"])
rs = d["rot_scale"]
print("RS", rs["value"], rs["ms_per_step"], rs["roofline"]["frac"], rs["parity"], {k: round(v["pairs_per_s"], 1) for k, v in rs["cpu_baseline"]["by_threads"].items()}, rs["gpu_vs_cpu"])
for k in ("orb", "sift"):
    e = d["descriptors_to_filtered_matches"][k]
    print(k, e["value"], e["matcher_ms_per_step"], e["roofline"]["frac"], e["roofline"]["achieved"])
for k in ("rot_scale_bench_512_per_launch", "rot_scale_bench_8192_per_launch", "rot_scale_bench_512_per_launch_GMS_SCALE_PROBE_0"):
    print(k, {a: (round(b["pairs_per_s"]), round(b["ms_per_launch"], 4), b["mismatches"]) for a, b in misc[k].items()})
for k, v in misc.items():
    if k.startswith(("config", "one_shot", "per_pixel", "host_batch", "batch")):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
print({k: round(v["pairs_per_s"] / 1e6, 2) for k, v in misc.items() if k.startswith("stagger")})
print(misc["crowded_bench"])
x = json.loads(open(F + "bench_spatial.json").read().strip().splitlines()[-1]); print("spatial", x["value"], x["roofline"]["kernel_ms_per_launch"])
x = json.loads(open(F + "bench_2ranks.json").read().strip().splitlines()[-1]); print("2ranks", x["value"], x["parity"]["pairs_checked"])
for f in ("phase.json", "phase_rs_noprobe.json", "phase_rs.json"):
    x = json.load(open(F + f))
    for k, v in x.items():
        if isinstance(v, dict):
            print(f, k, round(v["total_cycles"]), {a: round(b["cycles"]) for a, b in v.get("phases", {}).items()})
for k in ("orb", "sift"):
    r = {a: float(b) for a, b, _ in list(csv.reader(open(f"profiles/{tag}_matcher_counters_{k}.csv")))[1:]}
    print(k, "mfma util", round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (r["GRBM_GUI_ACTIVE"] / 8), 3), "clock", round(r["GRBM_GUI_ACTIVE"] / 8 / r["kernel_duration_ns"], 3))
t = json.load(open(f"profiles/{tag}_pmc_traffic.json"))
print("traffic", t["hbm_read_bytes_per_launch"] / 1e9, t["hbm_write_bytes_per_launch"] / 1e9, t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"])
