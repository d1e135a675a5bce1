#!/usr/bin/env python3
"""Diagnostic (not product): phases of stream_filter_kernel on libgms_hip_diag.so (-DGMS_PHASE_TIMING) for a batch of large pairs with
rotation + scale hypotheses: mean shader-clock cycles of thread 0 per phase, by item class (scale, band). Run on the GPU box after
`make -C sfm-gms_amd/csrc libgms_hip_diag.so`: python tools/stream_phase_timing.py [matches=50000] [pairs=64]"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
capi = importlib.import_module("sfm-gms_amd.capi")
capi.library_path = lambda: os.path.join(ROOT, "sfm-gms_amd", "csrc", os.environ.get("GMS_DIAG_LIB", "libgms_hip_diag.so"))
import measure_misc as mm  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = mm.pkg.load_library()
ctx = mm.pkg.GmsContext(0)
items = 44
dbuf = torch.zeros(n * items * 8 + 64, dtype=torch.int64, device="cuda:0")
lib.gms_diag_set_buffer.argtypes = [C.c_void_p]
lib.gms_diag_set_buffer(dbuf.data_ptr())
res = mm.device_batch(ctx, m, n, True, True, reps=2)
torch.cuda.synchronize()
raw = dbuf.cpu().numpy()[: n * items * 8].reshape(-1, 8).astype(np.float64)
L = np.arange(n * items)
n8 = n & ~7
item = np.where(L < n8 * items, (L >> 3) % items, (L - n8 * items) % items)
names = ["start (pair, flags, row counts)", "clear + barrier", "bin + barrier", "verify + barrier", "table", "pooling", "coarser verify + table"]
classes = {"20x20 -> 10x10 (4 per pair)": item < 4, "28x28 -> 14x14 band (12)": (item >= 4) & (item < 16), "40x40 band (28)": item >= 16}
out = {"matches": m, "pairs": n, "pairs_per_s": res["pairs_per_s"], "ms_per_launch": res["ms_per_launch"], "by_item_class": {}}
tot_all = 0.0
for cname, sel in classes.items():
    mean = raw[sel].mean(axis=0)
    out["by_item_class"][cname] = {"total_cycles": float(mean[:7].sum()), "phases": {nm: float(v) for nm, v in zip(names, mean[:7])}}
    tot_all += float(raw[sel][:, :7].sum())
out["sum_of_workgroup_cycles_per_pair"] = tot_all / n
print(json.dumps(out, indent=1))
