#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md / INTEGRATION.md (never the headline metric). Run on the GPU box:
    python tools/measure_misc.py            everything, JSON on stdout
    python tools/measure_misc.py batch M N ROT SCALE    one device-resident batch measurement (used by the stagger A/B below)
PCIe-inclusive numbers are labelled as such."""
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("sfm-gms_amd")
synth = importlib.import_module("sfm-gms_amd.synth")
batch = importlib.import_module("sfm-gms_amd.batch")
dist = importlib.import_module("sfm-gms_amd.dist")


def timeit(fn, reps):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t))


def device_batch(ctx, n_kp, n_pairs, rot, scale, reps=5):
    """pairs/s of gms_filter_device on n_pairs resident pairs of n_kp matches each (matches generated on the device)."""
    import torch
    size = (1920, 1080) if n_kp <= 20000 else (3840, 2160)
    n_frames = 8
    while n_frames * (n_frames - 1) // 2 < n_pairs:
        n_frames += 8
    frames = synth.make_sequence(1000, n_frames, size=size, n_kp=n_kp)
    table = batch.FrameTable(ctx, frames, [size] * n_frames)
    dev = table.device
    pairs = dist.pair_table(n_frames, 0, n_pairs, n_kp)
    d_pairs = batch._to_dev(pairs, dev)
    d_matches = dist.synth_matches_device(0, n_pairs, n_kp, 0.5, dev)
    d_out = torch.zeros((n_pairs * n_kp, 4), dtype=torch.int32, device=dev)
    d_res = torch.zeros((n_pairs, 4), dtype=torch.int32, device=dev)
    ctx.reserve(n_pairs, n_kp, rot, scale)
    torch.cuda.synchronize()

    def run():
        ctx.filter_device(table.d_pts.data_ptr(), table.d_frame_off.data_ptr(), n_frames, d_pairs.data_ptr(), n_pairs, n_kp,
                          d_matches.data_ptr(), d_out.data_ptr(), d_res.data_ptr(), None, rot, scale, 6.0)
        ctx.synchronize()
    dt = timeit(run, reps)
    return {"pairs_per_s": n_pairs / dt, "ms_per_launch": dt * 1e3, "gmatches_per_s": n_pairs * n_kp / dt / 1e9}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "batch":
        m, n, rot, scale = int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4])), bool(int(sys.argv[5]))
        print(json.dumps(device_batch(pkg.GmsContext(0), m, n, rot, scale)))
        return
    import cases
    import gms_oracle
    ctx = pkg.GmsContext(0)
    out = {}
    # ---- the one-shot host-pointer call (PCIe inclusive): BASELINE configs 1, 2, 4 and the reference's per-pixel call sites
    for name, c, reps in (("config1_500", cases.random_pair(11, n=500, size1=(640, 480), inlier_frac=0.6), 50),
                          ("one_shot_10k", cases.random_pair(100, n=10000, inlier_frac=0.5), 30),
                          ("config4_50k", cases.random_pair(104, n=50000, size1=(3840, 2160), inlier_frac=0.5), 8)):
        for flags in ((False, False), (True, True)):
            gpu = timeit(lambda: ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), reps)
            cpu = timeit(lambda: gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags), 3)
            out[f"{name}_rot{int(flags[0])}_scale{int(flags[1])}"] = {
                "gpu_call_ms_incl_pcie": gpu * 1e3, "gpu_pairs_per_s": 1 / gpu, "cpu_oracle_ms_1thread": cpu * 1e3}
    import test_gpu_band_path as big
    c = big._per_pixel_case(2594, 1131, 4)
    gpu = timeit(lambda: ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"]), 3)
    cpu = timeit(lambda: gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"]), 2)
    out["per_pixel_2594x1131_2.93M_default_flags"] = {"gpu_call_ms_incl_pcie": gpu * 1e3, "cpu_oracle_ms_1thread": cpu * 1e3}
    # ---- the host-pointer batch entry (pinned staging, two streams): 2048 pairs x 10k matches from host memory
    size, n_frames, n_kp, n_pairs = (1920, 1080), 72, 10000, 2048
    frames = synth.make_sequence(1000, n_frames, size=size, n_kp=n_kp)
    pairs = dist.pair_table(n_frames, 0, n_pairs, n_kp)
    matches = np.concatenate([dist.synth_matches_host(k, n_kp, 0.5) for k in range(n_pairs)])
    for flags in ((False, False), (True, True)):
        dt = timeit(lambda: ctx.filter_host_batch(frames, [size] * n_frames, pairs, matches, *flags), 3)
        out[f"host_batch_2048x10k_rot{int(flags[0])}_scale{int(flags[1])}"] = {
            "pairs_per_s_incl_pcie": n_pairs / dt, "ms": dt * 1e3, "GB_per_s_host_to_device": n_pairs * n_kp * 16 / dt / 1e9}
    # ---- device-resident batches of large pairs
    out["batch256_50000_default_flags"] = device_batch(ctx, 50000, 256, False, False)
    out["batch256_168750_default_flags"] = device_batch(ctx, 168750, 256, False, False)
    out["batch64_50000_rot_scale"] = device_batch(ctx, 50000, 64, True, True, reps=3)
    # ---- the first-round stagger at other pair sizes (GMS_STAGGER_US is read once per process: one process per setting)
    for m in (500, 4000, 10000, 16384):
        for us in ("default", "0"):
            env = dict(os.environ)
            if us != "default":
                env["GMS_STAGGER_US"] = us
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "batch", str(m), "4096", "0", "0"], capture_output=True, text=True, env=env)
            out[f"stagger_{us}_m{m}_4096pairs"] = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
