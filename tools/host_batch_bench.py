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
kp_all = np.concatenate(frames)
frame_off = np.arange(n_frames + 1, dtype=np.int64) * n_kp
for flags in ((False, False), (True, True)):
    o, r = ctx.filter_host_batch((kp_all, frame_off), [size] * n_frames, pairs, matches, *flags)   # the arrays every later call writes into
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.filter_host_batch((kp_all, frame_off), [size] * n_frames, pairs, matches, *flags, out=o, results=r); t.append(time.perf_counter() - t0)
    dt = float(np.median(t))
    kept = int(r["n_inliers"].astype(np.int64).sum())
    out[f"host_batch_{n_pairs}x10k_rot{int(flags[0])}_scale{int(flags[1])}"] = {
        "pairs_per_s_incl_pcie": n_pairs / dt, "ms": dt * 1e3, "GB_per_s_in": n_pairs * n_kp * 16 / dt / 1e9, "GB_per_s_out": kept * 16 / dt / 1e9,
        "through": "the Python binding (ctypes) on reused output arrays; tools/ubench/host_batch_rate.cpp is the same call from C++"}
print(json.dumps(out))
