#!/usr/bin/env python3
"""Diagnostic: bench.py's descriptors_to_poses leg alone, on each library given (one process each):
    python tools/poses_bench.py [libgms_hip.so libgms_hip_base.so ...]"""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    capi = importlib.import_module("sfm-gms_amd.capi")
    capi.library_path = lambda: sys.argv[2]
    import bench
    pkg = importlib.import_module("sfm-gms_amd")
    dev = torch.device("cuda", 0)
    ctx = pkg.GmsContext(0)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    r = bench.poses_leg(ctx, pkg, stream, dev)
    print(json.dumps({"lib": os.path.basename(sys.argv[2]), "pairs_per_s": r["value"], "two_view_ms": r["two_view_ms_per_step"], "matcher_ms": r["matcher_ms_per_step"],
                      "poses": r["poses_found"], "parity": r["parity"]["ok"], "mean_ransac_iters": r["mean_ransac_iters"]}))
    sys.exit(0)
for lib in (sys.argv[1:] or ["libgms_hip.so"]):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", os.path.join(ROOT, "sfm-gms_amd", "csrc", lib)], capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("failed: " + r.stderr[-400:]))
