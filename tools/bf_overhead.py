#!/usr/bin/env python3
"""Diagnostic: what a matcher workgroup costs beyond its matrix instructions. Every pair has 10k query rows (40 workgroups of 256
queries); the train frame has 5k, 10k, 20k or 40k rows. time(n) = fixed + slope * n: the fixed part is launch + prologue + epilogue
of a workgroup, the slope the steady-state rate. Run on the GPU box: python tools/bf_overhead.py [orb|sift]"""
import importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sfm-gms_amd")
synth = importlib.import_module("sfm-gms_amd.synth")
batch = importlib.import_module("sfm-gms_amd.batch")
kind = sys.argv[1] if len(sys.argv) > 1 else "orb"
n_pairs = 512
sizes = [10000] * 4 + [5000] * 4 + [10000] * 4 + [20000] * 4 + [40000] * 4      # frames 0..3 = queries
rng = np.random.default_rng(1)
if kind == "orb":
    descs = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in sizes]
else:
    descs = [np.clip(np.rint(rng.gamma(1.2, 22.0, (n, 128))), 0, 255).astype(np.float32) for n in sizes]
ctx = pkg.GmsContext(0)
size = (1920, 1080)
frames = [synth.make_keypoints(np.stack([rng.uniform(0, size[0] - 1, n), rng.uniform(0, size[1] - 1, n)], axis=1)) for n in sizes]
table = batch.FrameTable(ctx, frames, [size] * len(frames))
dt = batch.DescriptorTable(ctx, table, descs, pkg.GMS_DESC_HAMMING256 if kind == "orb" else pkg.GMS_DESC_L2_F32X128)
dev = table.device
d_matches = torch.zeros((n_pairs * 10000, 4), dtype=torch.int32, device=dev)
out = {}
for g, n_train in enumerate([5000, 10000, 20000, 40000]):
    pairs = np.zeros(n_pairs, dtype=pkg.PAIR_DTYPE)
    k = np.arange(n_pairs)
    pairs["frame_a"], pairs["frame_b"], pairs["m"], pairs["match_off"] = k % 4, 4 + 4 * g + (k // 4) % 4, 10000, k * 10000
    d_pairs = batch._to_dev(pairs, dev)
    dt.match_device(d_pairs.data_ptr(), n_pairs, 10000, d_matches.data_ptr(), True)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        dt.match_device(d_pairs.data_ptr(), n_pairs, 10000, d_matches.data_ptr(), True)
    ctx.synchronize()
    out[n_train] = (time.perf_counter() - t0) / reps * 1e3
ms = out
slope = (ms[40000] - ms[10000]) / 30000
fixed = ms[10000] - slope * 10000
print(json.dumps({"kind": kind, "ms_per_512_pairs": ms, "slope_ms_per_1000_train_rows": slope * 1000, "fixed_ms": fixed,
                  "fixed_share_at_10k": fixed / ms[10000]}))
