#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the headline metric): the one-shot host-pointer call (PCIe inclusive),
and the large-pair kernel on BASELINE config 4. Run on the GPU box: python tools/measure_misc.py"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("sfm-gms_amd")
import cases  # noqa: E402
import gms_oracle  # noqa: E402

ctx = pkg.GmsContext(0)
out = {}


def timeit(fn, reps):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t))


c = cases.random_pair(100, n=10000, inlier_frac=0.5)
for flags in ((False, False), (True, True)):
    gpu = timeit(lambda: ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), 20)
    cpu = timeit(lambda: gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), 3)
    out[f"one_shot_10k_rot{int(flags[0])}_scale{int(flags[1])}"] = {
        "gpu_call_ms_incl_pcie": gpu * 1e3, "gpu_pairs_per_s": 1 / gpu, "cpu_oracle_ms_1thread": cpu * 1e3}
c = cases.random_pair(11, n=500, size1=(640, 480), inlier_frac=0.6)  # BASELINE config 1
for flags in ((False, False), (True, True)):
    gpu = timeit(lambda: ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), 50)
    cpu = timeit(lambda: gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), 5)
    out[f"config1_500_rot{int(flags[0])}_scale{int(flags[1])}"] = {
        "gpu_call_ms_incl_pcie": gpu * 1e3, "cpu_oracle_ms_1thread": cpu * 1e3}
c = cases.random_pair(104, n=50000, size1=(3840, 2160), inlier_frac=0.5)
for flags in ((False, False), (True, True)):
    gpu = timeit(lambda: ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), 5)
    cpu = timeit(lambda: gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), 2)
    out[f"config4_50k_rot{int(flags[0])}_scale{int(flags[1])}"] = {
        "gpu_call_ms_incl_pcie": gpu * 1e3, "cpu_oracle_ms_1thread": cpu * 1e3}
# device-resident batches of large pairs (256 pairs per launch, default flags): the band kernels, and the slab kernel alone
import subprocess
for feats in (50000, 168750):
    for band in ("1", "0"):
        env = dict(os.environ, GMS_BAND=band)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extra", "--features", str(feats),
                            "--pairs", "256", "--frames", "16", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, env=env)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out[f"batch256_{feats}_default_flags_band{band}"] = {"pairs_per_s": d["value"], "gmatches_per_s": d["value"] * feats / 1e9,
                                                              "ms_per_256_pairs": d["ms_per_step"]}
# the same with rotation + scale hypotheses (BASELINE config 4's flags): tiled LDS kernels vs the slab kernel alone, 64 pairs resident
code = """
import sys, json, importlib, argparse, torch
sys.path.insert(0, %r)
import bench
pkg = importlib.import_module("sfm-gms_amd"); synth = importlib.import_module("sfm-gms_amd.synth")
ctx = pkg.GmsContext(0); dev = torch.device("cuda", 0); stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream)
args = argparse.Namespace(pairs=64, frames=16, features=50000, inlier_frac=0.5)
wl = bench.build_workload(args, 0, 1, dev, pkg, synth, ctx)
w, k = bench.timed_steps(ctx, wl, stream, 3, 1, True, True, None)
print(json.dumps({"pairs_per_s": 64 * 3 / w, "ms_per_64_pairs": w / 3 * 1e3}))
""" % ROOT
for band in ("1", "0"):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GMS_BAND=band))
    out[f"batch64_50000_rot_scale_band{band}"] = json.loads(r.stdout.strip().splitlines()[-1])
print(json.dumps(out, indent=1))
