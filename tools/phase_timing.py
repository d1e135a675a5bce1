#!/usr/bin/env python3
"""Diagnostic (not product): run the bench workload on libgms_hip_diag.so (built with -DGMS_PHASE_TIMING)
and print the mean shader-clock cycles thread 0 of a workgroup spends in each phase of filter_kernel.
Read the SHARES, not the length: the stamps themselves cost cycles."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PHASES = ["bin_barrier", "region_tables", "clear", "insert_first", "bin_qt_loads", "verify", "mark", "count_select", "out_scan", "copy_out", "insert_leftover", "insert_barrier", "bin_gathers", "bin_compute", "x14", "x15"]


BY_SCALE = ["eval_s0", "eval_s1", "eval_s2", "eval_s3", "eval_s4", "probe_s0", "probe_s1", "probe_s2", "probe_s3", "probe_s4", "load", "record", "sort", "x13", "x14", "x15"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--rot", type=int, default=0)
    ap.add_argument("--scale", type=int, default=0)
    ap.add_argument("--features", type=int, default=10000)
    ap.add_argument("--lib", default="libgms_hip_diag.so", help="diagnostic library under sfm-gms_amd/csrc (libgms_hip_diag_byscale.so: built with -DGMS_STAMP_BY_SCALE)")
    ap.add_argument("--by-scale", action="store_true", help="label the sums of the scale-hypothesis kernel as a -DGMS_STAMP_BY_SCALE build writes them")
    ap.add_argument("--starts", default="", help="save per-pair (start, records landed) wall-clock stamps to this .npy")
    a = ap.parse_args()
    capi = importlib.import_module("sfm-gms_amd.capi")
    diag_path = os.path.join(ROOT, "sfm-gms_amd", "csrc", a.lib)
    capi.library_path = lambda: diag_path
    pkg = importlib.import_module("sfm-gms_amd")
    lib = pkg.load_library()
    dev = torch.device("cuda", 0)
    ctx = pkg.GmsContext(0)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    args = argparse.Namespace(pairs=a.pairs, frames=200, features=a.features, inlier_frac=0.5, warmup=1, steps=1, max_resident=2)
    wl = bench.Workload(args, 0, 1, dev, pkg, ctx)
    dbuf = torch.zeros(2 * a.pairs * 16, dtype=torch.int64, device=dev)  # with scale hypotheses: first kernel's stamps, then the second's
    lib.gms_diag_set_buffer.argtypes = [C.c_void_p]
    lib.gms_diag_set_buffer(dbuf.data_ptr())
    for _ in range(3):
        with torch.cuda.stream(stream):
            wl.launch(ctx, 0, bool(a.rot), bool(a.scale))
    torch.cuda.synchronize()
    raw_all = dbuf.cpu().numpy().reshape(2, -1, 16)
    out = {"pairs": a.pairs, "rot": a.rot, "scale": a.scale}
    for name, raw in (("kernel_1", raw_all[0]), ("kernel_2_scale4_hashed", raw_all[1])):
        if name != "kernel_1" and not raw.any():
            continue
        if a.starts and name == "kernel_1":
            np.save(a.starts, raw[:, 14:16])
        d = raw[:, :16].astype(np.float64)
        d[:, 14:16] = 0
        mean = d.mean(axis=0)
        tot = mean.sum()
        names = BY_SCALE if (a.by_scale and name == "kernel_1") else PHASES
        out[name] = {"total_cycles": tot, "phases": {n: {"cycles": float(c), "share": float(c / tot)} for n, c in zip(names, mean) if c > 0}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
