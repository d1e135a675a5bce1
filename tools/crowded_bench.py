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


def squeezed(case_id, n_frames, size=(1920, 1080), n_kp=10000, drift_px=6.0, noise_px=1.5, spatial_order=False):
    frames = orig(case_id, n_frames, size=size, n_kp=n_kp, drift_px=drift_px, noise_px=noise_px, spatial_order=spatial_order)
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
    args = argparse.Namespace(pairs=4096, frames=200, features=10000, inlier_frac=0.5, warmup=2, steps=10, max_resident=12)
    wl = bench.Workload(args, 0, 1, dev, pkg, ctx)
    w, k = bench.timed_steps(ctx, wl, stream, 10, 2, False, False, None)
    c, b = bench.check_parity(wl, pkg, range(len(wl.chunks)), False, False, sample={i: list(range(0, 4096, 512)) for i in range(len(wl.chunks))})
    out[name] = {"pairs_per_s": 4096 * 10 / w, "kernel_ms": k, "parity_pairs": c, "mismatches": b}
    w2, k2 = bench.timed_steps(ctx, wl, stream, 6, 2, True, True, None, n_pairs=512)
    used = sorted({(2 + s) % len(wl.chunks) for s in range(6)})
    c2, b2 = bench.check_parity(wl, pkg, used, True, True, sample={i: list(range(0, 512, 128)) for i in used})
    out[name]["rot_scale_pairs_per_s"] = 512 * 6 / w2
    out[name]["rot_scale_parity_pairs"] = c2
    out[name]["rot_scale_mismatches"] = b2
    ctx.close()
print(json.dumps(out, indent=1))
