#!/usr/bin/env python3
"""After tools/profile.sh, tools/pmc_collect.sh and tools/final_side.sh have written gpurun_out/: copy what is to be judged into
profiles/ (through tools/make_profiles.py) and print the numbers the documents quote. python tools/collect_final.py [r03]"""
import csv, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
os.chdir(ROOT)
subprocess.check_call([sys.executable, "tools/make_profiles.py", tag])
F = "gpurun_out/final/"
for src, dst in (("bench.json", "bench.json"), ("phase.json", "phase_cycles.json"), ("phase_rs.json", "phase_cycles_rot_scale.json"),
                 ("phase_rs_noprobe.json", "phase_cycles_rot_scale_probe_off.json"), ("stream_phase.json", "stream_phase_cycles.json"),
                 ("config4_bench.json", "config4_bench.json"), ("kernel_stats_rot_scale_64.csv", "kernel_stats_config4_rot_scale.csv"),
                 ("kernel_stats_default_256.csv", "kernel_stats_config4_default_flags.csv"), ("matcher_limits.txt", "matcher_limits.txt"),
                 ("matcher_counters_orb.csv", "matcher_counters_orb.csv"), ("matcher_counters_sift.csv", "matcher_counters_sift.csv"),
                 ("image_pair_sparse.json", "image_pair_sparse.json"), ("image_pair_dense.json", "image_pair_dense.json"),
                 ("host_copy_rate.json", "host_copy_rate.json"), ("host_batch_rate.json", "host_batch_rate.json"), ("host_batch_python.json", "host_batch_python.json"),
                 ("real_pixels.json", "real_pixels.json"), ("real_pixels_list_order.json", "real_pixels_list_order.json"), ("ab_scales.txt", "ab_scales_probe_forms.txt")):
    shutil.copy(F + src, f"profiles/{tag}_{dst}")
for name in ("kernel_stats_config4_rot_scale.csv", "kernel_stats_config4_default_flags.csv"):   # kernel names cut, as in the main stats files
    rows = list(csv.DictReader(open(f"profiles/{tag}_{name}")))
    with open(f"profiles/{tag}_{name}", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 tools/measure_misc.py batch 50000 N rot scale  (tools/config4_prof.sh; names cut at 110 chars)\n")
        w = csv.writer(f)
        w.writerow(list(rows[0].keys()))
        for r in rows:
            w.writerow([r[k][:110] if k == "Name" else r[k] for k in rows[0].keys()])
misc = json.load(open(F + "misc.json"))
misc["crowded_bench"] = json.load(open(F + "crowded.json"))
json.dump(misc, open(f"profiles/{tag}_side_measurements.json", "w"), indent=1)
d = json.loads(open(F + "bench.json").read().strip().splitlines()[-1])
print("HEAD", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["achieved"])
print("CPU", {k: round(v["pairs_per_s"]) for k, v in d["cpu_baseline"]["by_threads"].items()}, d["gpu_vs_cpu"])
rs = d["rot_scale"]
print("RS", rs["value"], rs["ms_per_step"], rs["roofline"]["frac"], rs["parity"], rs["by_probe"], rs["gpu_vs_cpu"])
for k in ("orb", "sift"):
    e = d["descriptors_to_filtered_matches"][k]
    print(k, e["value"], e["matcher_ms_per_step"], e["roofline"]["frac"], e["roofline"]["achieved"])
for k, v in misc.items():
    if k.startswith(("config", "one_shot", "per_pixel", "host_batch", "batch")):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
print({k: round(v["pairs_per_s"] / 1e6, 2) for k, v in misc.items() if k.startswith("stagger")})
print(misc["crowded_bench"])
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
