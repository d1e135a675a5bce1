#!/usr/bin/env python3
"""Diagnostic: build libgms_hip.so variants with extra -D flags on the GPU box and time the bench workload on each."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "sfm-gms_amd", "csrc")
for i, flags in enumerate(sys.argv[1:]):
    out = f"/tmp/libgms_var{i}.so"
    fl = [f for f in flags.split() if f != "none"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w", "-mllvm", "-disable-machine-licm",
                           "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-shared", "-o", out,
                           os.path.join(csrc, "gms_kernels.hip"),
                           os.path.join(csrc, "gms_kernel_big.hip"), os.path.join(csrc, "gms_kernel_band.hip"), os.path.join(csrc, "gms_capi.cpp")] + fl)
    code = f"""
import importlib, sys, json
sys.path.insert(0, {ROOT!r}); sys.argv = ['bench.py'] + {os.environ.get('BENCH_ARGS', '--steps 10 --warmup 2 --no-extra --no-cpu').split()!r}
capi = importlib.import_module('sfm-gms_amd.capi'); capi.library_path = lambda: {out!r}
import bench; bench.main()
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("VARIANT", flags, round(d["roofline"]["kernel_ms_per_launch"], 4), "ms/kernel", round(d["ms_per_step"], 4), "ms/step", "parity", d.get("parity", {}).get("bit_exact"))
    except Exception:
        print("VARIANT", flags, "failed", r.stderr[-300:])
