#!/usr/bin/env python3
"""Diagnostic: gms_filter_host_batch alone (2048 pairs x 10k matches from pageable host memory), PCIe inclusive."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sfm-gms_amd"); synth = importlib.import_module("sfm-gms_amd.synth"); dist = importlib.import_module("sfm-gms_amd.dist")
ctx = pkg.GmsContext(0)
size, n_frames, n_kp, n_pairs = (1920, 1080), 72, 10000, 2048
frames = synth.make_sequence(1000, n_frames, size=size, n_kp=n_kp)
pairs = dist.pair_table(n_frames, 0, n_pairs, n_kp)
matches = np.concatenate([dist.synth_matches_host(k, n_kp, 0.5) for k in range(n_pairs)])
out = {}
for flags in ((False, False), (True, True)):
    ctx.filter_host_batch(frames, [size] * n_frames, pairs, matches, *flags)
    t = []
    for _ in range(4):
        t0 = time.perf_counter(); ctx.filter_host_batch(frames, [size] * n_frames, pairs, matches, *flags); t.append(time.perf_counter() - t0)
    dt = float(np.median(t))
    out[f"rot{int(flags[0])}_scale{int(flags[1])}"] = {"pairs_per_s_incl_pcie": n_pairs / dt, "ms": dt * 1e3, "GB_per_s_each_way": n_pairs * n_kp * 16 / dt / 1e9}
print(json.dumps(out))
