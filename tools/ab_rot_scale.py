#!/usr/bin/env python3
"""Diagnostic A/B of the scale-hypothesis kernels: tools/rot_scale_bench.py on libgms_hip_base.so and libgms_hip.so in one session."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import importlib, sys
sys.path.insert(0, {root!r}); sys.argv = ['rot_scale_bench.py', {n!r}]
capi = importlib.import_module('sfm-gms_amd.capi'); capi.library_path = lambda: {lib!r}
__file__ = {root!r} + '/tools/rot_scale_bench.py'
exec(open(__file__).read())
"""
n = sys.argv[1] if len(sys.argv) > 1 else "512"
for rnd in range(2):
    for lib in ("libgms_hip_base.so", "libgms_hip.so"):
        r = subprocess.run([sys.executable, "-c", code.format(root=ROOT, n=n, lib=ROOT + "/sfm-gms_amd/csrc/" + lib)], capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(lib, {k: (round(v["pairs_per_s"]), v["mismatches"]) for k, v in d.items()})
        except Exception:
            print(lib, "failed", r.stderr[-400:])
