#!/usr/bin/env python3
"""Diagnostic A/B of the matcher: tools/bf_bench.py on libgms_hip_base.so and libgms_hip.so in one session on one device."""
import subprocess, sys, os
ROOT='/root/repo'
code = """
import importlib, sys
sys.path.insert(0, {root!r}); sys.argv = ['bf_bench.py', {kind!r}, '1024']
capi = importlib.import_module('sfm-gms_amd.capi'); capi.library_path = lambda: {lib!r}
__file__ = {root!r} + '/tools/bf_bench.py'
exec(open(__file__).read())
"""
for rnd in range(2):
    for lib in (sys.argv[1:] or ['libgms_hip_base.so','libgms_hip.so']):
        for kind in ('orb','sift'):
            r = subprocess.run([sys.executable, '-c', code.format(root=ROOT, kind=kind, lib=ROOT+'/sfm-gms_amd/csrc/'+lib)], capture_output=True, text=True)
            import json
            try:
                d=json.loads(r.stdout.strip().splitlines()[-1]); print(lib, kind, round(d[kind]['match']['pairs_per_s']))
            except Exception: print(lib, kind, 'failed', r.stderr[-300:])
