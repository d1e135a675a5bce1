#!/usr/bin/env python3
"""Diagnostic: throughput of the brute-force matcher (gms_bfmatch_device) on 10k x 10k frames, and of matcher + GMS filter
back to back. Run on the GPU box: python tools/bf_bench.py [orb|sift] [pairs]"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sfm-gms_amd")
synth = importlib.import_module("sfm-gms_amd.synth")
batch = importlib.import_module("sfm-gms_amd.batch")
d = importlib.import_module("sfm-gms_amd.dist")

kind = sys.argv[1] if len(sys.argv) > 1 else "orb"
use_prepared = "valu" not in kind
kind = kind.replace("-valu", "")
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_kp, size = 10000, (1920, 1080)
n_frames = 32
while n_frames * (n_frames - 1) // 2 < n_pairs:
    n_frames += 8
ctx = pkg.GmsContext(0)
frames = synth.make_sequence(1000, n_frames, size=size, n_kp=n_kp)
descs = synth.sequence_descriptors(1000, n_frames, n_kp, kind, outlier_frac=0.5)
if os.environ.get("BF_ZERO") == "1":   # diagnostic: all-zero rows draw less power -- how far the clock under load limits the kernel
    descs = [np.zeros_like(x) for x in descs]
table = batch.FrameTable(ctx, frames, [size] * n_frames)
dt = batch.DescriptorTable(ctx, table, descs, pkg.GMS_DESC_HAMMING256 if kind == "orb" else pkg.GMS_DESC_L2_F32X128)
pairs = d.pair_table(n_frames, 0, n_pairs, n_kp)
dev = table.device
d_pairs = batch._to_dev(pairs, dev)
d_matches = torch.zeros((n_pairs * n_kp, 4), dtype=torch.int32, device=dev)
d_out = torch.zeros((n_pairs * n_kp, 4), dtype=torch.int32, device=dev)
d_res = torch.zeros((n_pairs, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def run(with_filter):
    dt.match_device(d_pairs.data_ptr(), n_pairs, n_kp, d_matches.data_ptr(), use_prepared)
    if with_filter:
        ctx.filter_device(table.d_pts.data_ptr(), table.d_frame_off.data_ptr(), n_frames, d_pairs.data_ptr(), n_pairs, n_kp,
                          d_matches.data_ptr(), d_out.data_ptr(), d_res.data_ptr(), None, False, False, 6.0)


out = {}
for with_filter in (False, True):
    run(with_filter)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        run(with_filter)
    ctx.synchronize()
    dtm = (time.perf_counter() - t0) / reps
    ops = n_pairs * n_kp * n_kp * (128 * 2 if kind == "sift" else 1)
    out["match+filter" if with_filter else "match"] = {
        "ms": dtm * 1e3, "pairs_per_s": n_pairs / dtm,
        ("TFLOP/s" if kind == "sift" else "G distance evaluations/s"): ops / dtm / (1e12 if kind == "sift" else 1e9)}
res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(-1)
out["mean_kept"] = float(res["n_inliers"].mean())
out["true_match_rate"] = float((d_matches[:n_kp, 1] == d_matches[:n_kp, 0]).float().mean().item())
print(json.dumps({kind: out}))
