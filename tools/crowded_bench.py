#!/usr/bin/env python3
"""Diagnostic: the headline workload with the keypoints squeezed into the central part of the image, so that every populated
left cell holds more than 255 matches and every pair leaves the byte-matrix path for the hashed one inside the same kernel."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("sfm-gms_amd")
synth = importlib.import_module("sfm-gms_amd.synth")
import argparse  # noqa: E402

orig = synth.make_sequence


def squeezed(case_id, n_frames, size=(1920, 1080), n_kp=10000, drift_px=6.0, noise_px=1.5):
    frames = orig(case_id, n_frames, size=size, n_kp=n_kp, drift_px=drift_px, noise_px=noise_px)
    w, h = size
    for f in frames:  # shrink towards the centre by 2.5x: 0.7 * 0.7 of the image -> 0.28 * 0.28 (about 36 cells)
        f["x"] = (w / 2 + (f["x"] - w / 2) / 2.5).astype(f["x"].dtype)
        f["y"] = (h / 2 + (f["y"] - h / 2) / 2.5).astype(f["y"].dtype)
    return frames


out = {}
for name, fn in (("uniform", orig), ("crowded", squeezed)):
    synth.make_sequence = fn
    ctx = pkg.GmsContext(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    args = argparse.Namespace(pairs=4096, frames=128, features=10000, inlier_frac=0.5)
    wl = bench.build_workload(args, 0, 1, dev, pkg, synth, ctx)
    w, k = bench.timed_steps(ctx, wl, stream, 10, 2, False, False, None)
    rate_mt, rate1, ok, n = bench.cpu_leg(argparse.Namespace(cpu_pairs=64, cpu_threads=16), wl, pkg, False, False, 64, 16)
    out[name] = {"pairs_per_s": 4096 * 10 / w, "kernel_ms": k, "parity_64_pairs": ok}
    sub = dict(wl)
    sub["n_pairs"] = 512
    w2, k2 = bench.timed_steps(ctx, sub, stream, 6, 2, True, True, None)
    rate_mt2, rate12, ok2, n2 = bench.cpu_leg(argparse.Namespace(cpu_pairs=16, cpu_threads=16), sub, pkg, True, True, 16, 16)
    out[name]["rot_scale_pairs_per_s"] = 512 * 6 / w2
    out[name]["rot_scale_parity_16_pairs"] = ok2
    ctx.close()
print(json.dumps(out, indent=1))
