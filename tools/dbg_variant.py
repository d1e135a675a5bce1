#!/usr/bin/env python3
"""Diagnostic: build libgms_hip.so variants with extra -D flags on the GPU box and run a few parity cases."""
import importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gms_oracle
flags = sys.argv[1:]
csrc = os.path.join(ROOT, "sfm-gms_amd", "csrc")
out = "/tmp/libgms_dbg.so"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w",
       "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", out,
       os.path.join(csrc, "gms_kernels.hip"), os.path.join(csrc, "gms_capi.cpp")] + flags
subprocess.check_call(cmd)
capi = importlib.import_module("sfm-gms_amd.capi")
capi.library_path = lambda: out
pkg = importlib.import_module("sfm-gms_amd")
synth = importlib.import_module("sfm-gms_amd.synth")
ctx = pkg.GmsContext(0)
for name, size, n, case in [("cfg1", (640, 480), 500, 11), ("cfg2", (1920, 1080), 10000, 100)]:
    kp1, kp2, m = synth.make_pair(case, size1=size, n1=n, inlier_frac=0.6 if n == 500 else 0.5)
    for rot, sc in [(0, 0), (1, 1)]:
        got, res = ctx.match(size, size, kp1, kp2, m, rot, sc, 6.0, return_result=True)
        rc, want, mask, wres = gms_oracle.match(size, size, kp1, kp2, m, rot, sc, 6.0)
        print(flags, name, rot, sc, "gpu", len(got), tuple(res), "oracle", len(want), tuple(wres),
              "OK" if got.tobytes() == want.tobytes() else "MISMATCH")
