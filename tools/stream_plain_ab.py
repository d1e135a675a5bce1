#!/usr/bin/env python3
"""Diagnostic: large default-flags pairs on stream_plain_kernel (default) and, with GMS_STREAM_PLAIN=0 in a child process, on
stream_dense_kernel<false>: python tools/stream_plain_ab.py -- or on two builds of the library: python tools/stream_plain_ab.py libA.so libB.so"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CASES = [(50000, 256), (50000, 64), (20000, 256), (65000, 128)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    if len(sys.argv) > 2:
        import importlib
        sys.path.insert(0, ROOT)
        capi = importlib.import_module("sfm-gms_amd.capi")
        capi.library_path = lambda: os.path.abspath(sys.argv[2])
    import measure_misc as mm
    ctx = mm.pkg.GmsContext(0)
    print(json.dumps({f"batch{n}_{m}": mm.device_batch(ctx, m, n, False, False, reps=3) for m, n in CASES}))
    sys.exit(0)
libs = sys.argv[1:3]
legs = [(os.path.basename(l), {}, [l]) for l in libs] if len(libs) == 2 else [("plain", {}, []), ("dense", {"GMS_STREAM_PLAIN": "0"}, [])]
for rnd in range(2):
    for tag, env, extra in legs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + extra, capture_output=True, text=True, env=dict(os.environ, **env))
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(tag, {k: round(v["pairs_per_s"]) for k, v in d.items()}, flush=True)
        except Exception:
            print(tag, "failed", r.stderr[-400:], flush=True)
