#!/usr/bin/env python3
"""The reference's disparity demo (DisparityUtil.cpp:93-201, alg "GMS") from pixels, every stage on the GPU:
    two 8-bit grey images -> keypoints + 32-byte rows (gms_detect_batch_device; --dense: a keypoint per interior pixel, gms_describe_device,
    DisparityUtil.cpp:123-133) -> one match per left keypoint (gms_bfmatch_device, NORM_HAMMING) -> matchGMS (gms_filter_device)
    -> disparity map + RMS against a ground truth (gms_disparity_device).
Input: an .npz with arrays left, right [H, W] uint8 and optionally gt (default: the committed 450 x 375 pair of the reference's
SourceImages, tests/golden/image_stereo_pair_450x375.npz). Prints one JSON line; --out writes the survivors and the map as .npz.
    python tools/gms_image_pair.py [pair.npz] [--dense] [--threshold 12] [--max-keypoints 10000] [--rotation] [--scale] [--ratio 4] [--check]
--check runs the CPU statement (oracle/) beside it and compares every stage (test infrastructure; slow in --dense)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pair", nargs="?", default=os.path.join(ROOT, "tests", "golden", "image_stereo_pair_450x375.npz"))
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--threshold", type=int, default=12)
    ap.add_argument("--max-keypoints", type=int, default=10000)
    ap.add_argument("--rotation", action="store_true")
    ap.add_argument("--scale", action="store_true")
    ap.add_argument("--ratio", type=int, default=4)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out")
    a = ap.parse_args()
    pkg = importlib.import_module("sfm-gms_amd")
    batch = importlib.import_module("sfm-gms_amd.batch")
    types = importlib.import_module("sfm-gms_amd.types")
    z = np.load(a.pair)
    left, right = np.ascontiguousarray(z["left"], dtype=np.uint8), np.ascontiguousarray(z["right"], dtype=np.uint8)
    gt = np.ascontiguousarray(z["gt"], dtype=np.uint8) if "gt" in z.files else None
    h, w = left.shape
    assert right.shape == (h, w), "images of one size (as a stereo pair)"
    ctx = pkg.GmsContext(0)
    t = {}
    t0 = time.perf_counter()
    if a.dense:
        b = pkg.GMS_DETECT_BORDER
        xs, ys = np.meshgrid(np.arange(b, w - b), np.arange(b, h - b), indexing="ij")     # DisparityUtil.cpp:125-130: columns outer
        grid = np.zeros(xs.size, dtype=pkg.KEYPOINT_DTYPE)
        grid["x"], grid["y"], grid["size"] = xs.ravel(), ys.ravel(), 1.0
        kps, rows = [], []
        for img in (left, right):
            st, k, r = batch.describe_image(ctx, img, grid)
            assert st == 0
            kps.append(k)
            rows.append(r)
    else:
        kps, rows = batch.detect_images(ctx, np.stack([left, right]), a.threshold, a.max_keypoints)
    t["keypoints_ms"] = (time.perf_counter() - t0) * 1e3
    table = batch.FrameTable(ctx, kps, [(w, h)] * 2)
    dt = batch.DescriptorTable(ctx, table, rows, pkg.GMS_DESC_HAMMING256)
    pairs = np.zeros(1, dtype=pkg.PAIR_DTYPE)
    pairs[0] = (0, 1, len(kps[0]), 0, 0)
    t0 = time.perf_counter()
    matches = batch.match_pairs(ctx, dt, pairs)
    t["match_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, a.rotation, a.scale, 6.0)
    t["filter_ms"] = (time.perf_counter() - t0) * 1e3
    n = int(res["n_inliers"][0])
    dev = table.device
    d_matches = batch._to_dev(out[:max(n, 1)], dev)
    d_n = torch.tensor([n], dtype=torch.int32, device=dev)
    d_gt = torch.from_numpy(gt).to(dev) if gt is not None else None
    d_disp = torch.zeros(w * h, dtype=torch.uint8, device=dev)
    d_work = torch.zeros(w * h, dtype=torch.int32, device=dev)
    d_stats = torch.zeros(24, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.disparity_device(table.d_kp.data_ptr(), len(kps[0]), table.d_kp.data_ptr() + len(kps[0]) * 28, len(kps[1]), d_matches.data_ptr(),
                         d_n.data_ptr(), n, w, h, d_gt.data_ptr() if d_gt is not None else None, a.ratio, d_disp.data_ptr(), d_work.data_ptr(),
                         d_stats.data_ptr())
    ctx.synchronize()
    t["disparity_ms"] = (time.perf_counter() - t0) * 1e3
    stats = d_stats.cpu().numpy().view(types.DISPARITY_STATS_DTYPE)[0]
    disp = d_disp.cpu().numpy().reshape(h, w)
    cnt = int(stats["count"])
    line = {"image": [w, h], "mode": "dense" if a.dense else "sparse", "keypoints": [len(kps[0]), len(kps[1])], "matches": len(matches),
            "flags": [a.rotation, a.scale, 6.0], "survivors": n, "best_scale": int(res["best_scale"][0]), "best_rot": int(res["best_rot"][0]),
            "status": int(res["status"][0]), "rms_pixels_compared": cnt,
            "disparity_rms": float(np.sqrt(float(stats["sum_sq"]) / cnt)) if cnt else None, "max_abs_error": int(stats["max_abs"]),
            "ms": {k: round(v, 3) for k, v in t.items()}, "note": "first-call times (allocation and module load included)"}
    if a.check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import gms_oracle as oracle
        ok = {}
        if a.dense:
            ok["keypoints"] = all(oracle.describe(img, grid)[2].tobytes() == rows[i].tobytes() for i, img in enumerate((left, right)))
            sample = np.arange(0, len(kps[0]), max(len(kps[0]) // 400, 1))
            want_m = oracle.bf_match(rows[0][sample], rows[1], True)
            ok["matches_sampled"] = bool((want_m["trainIdx"] == matches["trainIdx"][sample]).all() and (want_m["distance"] == matches["distance"][sample]).all()
                                         and (matches["queryIdx"] == np.arange(len(matches))).all() and (matches["imgIdx"] == 0).all())
        else:
            want = [oracle.detect(img, a.threshold, a.max_keypoints) for img in (left, right)]
            ok["keypoints"] = all(want[i][0].tobytes() == kps[i].tobytes() and want[i][1].tobytes() == rows[i].tobytes() for i in range(2))
            ok["matches"] = oracle.bf_match(rows[0], rows[1], True).tobytes() == matches.tobytes()
        rc, wout, wmask, wres = oracle.match((w, h), (w, h), kps[0], kps[1], matches, a.rotation, a.scale, 6.0)
        ok["filter"] = rc == 0 and wout.tobytes() == out[:n].tobytes()
        rc, wdisp, wcnt, wssq, wmx, wrms = oracle.disparity(kps[0], kps[1], wout, w, h, gt, a.ratio)
        ok["disparity"] = bool(np.array_equal(wdisp, disp)) and (wcnt, wssq, wmx) == (cnt, int(stats["sum_sq"]), int(stats["max_abs"]))
        line["check_vs_oracle"] = ok
    if a.out:
        np.savez_compressed(a.out, survivors=out[:n], disparity=disp, keypoints_left=kps[0], keypoints_right=kps[1])
    print(json.dumps(line))
    return 0 if not a.check or all(line["check_vs_oracle"].values()) else 1


if __name__ == "__main__":
    sys.exit(main())
