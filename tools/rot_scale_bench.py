#!/usr/bin/env python3
"""Diagnostic: the rot+scale side measurement of bench.py alone (matchGMS(true, true, 6.0) on 512 pairs x 10k matches), with a
parity check of every 16th pair. Run on the GPU box: python tools/rot_scale_bench.py [pairs per launch]"""
import argparse
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("sfm-gms_amd")
dev = torch.device("cuda", 0)
ctx = pkg.GmsContext(0)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
args = argparse.Namespace(pairs=N, frames=200, features=10000, inlier_frac=0.5, warmup=2, steps=8, max_resident=10)
wl = bench.Workload(args, 0, 1, dev, pkg, ctx)
out = {}
for rot, scale in ((True, True), (True, False), (False, True)):
    w, k = bench.timed_steps(ctx, wl, stream, 8, 2, rot, scale, None)
    c, b = bench.check_parity(wl, pkg, range(len(wl.chunks)), rot, scale, sample={i: list(range(0, N, N // 32)) for i in range(len(wl.chunks))})
    out[f"rot{int(rot)}_scale{int(scale)}"] = {"pairs_per_s": N * 8 / w, "ms_per_launch": k, "pairs_per_launch": N, "parity_checked": c, "mismatches": b}
print(json.dumps(out))
