#!/usr/bin/env python3
"""Diagnostic: time of the five-point minimal solver on the GPU (gms_selftest_five_point: the solver exactly as the RANSAC kernel runs
it, sixteen samples per workgroup) on 16 384 samples of a synthetic two-view scene, for each library given (default: the product one).
Timing-only builds: -DTV_DIAG=1 no Gauss-Newton polish, =2 stop before the root finder, =3 stop behind it.
    python tools/fivepoint_bench.py [libgms_hip.so libgms_hip_tv1.so ...]"""
import importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    sys.path.insert(0, ROOT)
    capi = importlib.import_module("sfm-gms_amd.capi")
    capi.library_path = lambda: sys.argv[2]
    pkg = importlib.import_module("sfm-gms_amd")
    synth = importlib.import_module("sfm-gms_amd.synth")
    sc = synth.make_two_view_scene(3, n_points=4000)
    K = sc["camera"]
    uv1 = np.stack([sc["frames"][0]["x"], sc["frames"][0]["y"]], axis=1).astype(np.float64)
    uv2 = np.stack([sc["frames"][1]["x"], sc["frames"][1]["y"]], axis=1).astype(np.float64)
    x1 = (uv1 - [K[2], K[3]]) / [K[0], K[1]]
    x2 = (uv2 - [K[2], K[3]]) / [K[0], K[1]]
    rng = np.random.default_rng(1)
    n = 16384
    idx = np.stack([rng.choice(len(x1), 5, replace=False) for _ in range(n)])
    ctx = pkg.GmsContext(0)
    ctx.selftest_five_point(x1[idx[:64]], x2[idx[:64]])
    t0 = time.perf_counter()
    models = ctx.selftest_five_point(x1[idx], x2[idx])
    dt = time.perf_counter() - t0
    print(json.dumps({"lib": os.path.basename(sys.argv[2]), "ms_for_16384_samples": dt * 1e3, "mean_models_per_sample": float(np.mean([len(m) for m in models]))}))
    sys.exit(0)
libs = sys.argv[1:] or ["libgms_hip.so"]
for lib in libs:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", os.path.join(ROOT, "sfm-gms_amd", "csrc", lib)], capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("failed: " + r.stderr[-400:]))
