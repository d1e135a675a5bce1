#!/usr/bin/env python3
"""Diagnostic A/B: the headline bench on two builds of the library in ONE session on ONE device (timings from different boxes differ
by 1-3 %). python tools/ab_bench.py libA.so libB.so [bench args...]; alternates A, B, A, B."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
extra = sys.argv[3:]
code = """
import importlib, sys
sys.path.insert(0, {root!r}); sys.argv = ['bench.py', '--no-cpu', '--no-extra'] + {extra!r}
capi = importlib.import_module('sfm-gms_amd.capi'); capi.library_path = lambda: {lib!r}
import bench; bench.main()
"""
for rnd in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", code.format(root=ROOT, extra=extra, lib=os.path.abspath(lib))], capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(os.path.basename(lib), round(d["roofline"]["kernel_ms_per_launch"], 4), "ms/launch", round(d["value"]), "pairs/s", d["parity"]["bit_exact"])
        except Exception:
            print(os.path.basename(lib), "failed", r.stderr[-300:])
