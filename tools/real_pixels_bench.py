#!/usr/bin/env python3
"""Diagnostic: bench.py's real_pixels leg alone (the reference's main() scenario from the committed 1080p fixture). Prints one JSON line."""
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
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
print(json.dumps(bench.real_pixels_leg(ctx, pkg, stream, dev, copies=copies)))
