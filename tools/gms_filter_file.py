#!/usr/bin/env python3
"""tools/gms_filter_file.py -- a GMSFRM01 dataset file (include/gms.h "ingest format") to filtered matches on the GPU:

    python tools/gms_filter_file.py seq.gmsf [--rot] [--scale] [--thr 6.0] [--match] [--camera fx fy cx cy] [--dist k1 k2 p1 p2 k3]
                                             [--prob 0.7] [--ransac-threshold 1.0] [--out result.npz]

The file is read by the library's C reader (gms_dataset_read); with descriptors and no matches in it (or --match) the putative
matches come from gms_bfmatch_device (FeatureMatchUtil.cpp:66-68), then gms_filter_device (matchGMS, FeatureMatchUtil.cpp:69), and
with --camera the two-view stage of structureFromMotion (SfMUtil.cpp:25-82: findEssentialMat, recoverPose, undistort + triangulate).
Prints one JSON line; --out keeps every array (numpy .npz)."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--rot", action="store_true")
    ap.add_argument("--scale", action="store_true")
    ap.add_argument("--thr", type=float, default=6.0)
    ap.add_argument("--match", action="store_true", help="brute-force match the descriptors even if the file holds matches")
    ap.add_argument("--camera", type=float, nargs=4, metavar=("FX", "FY", "CX", "CY"))
    ap.add_argument("--dist", type=float, nargs=5, metavar=("K1", "K2", "P1", "P2", "K3"))
    ap.add_argument("--prob", type=float, default=0.7, help="findEssentialMat's confidence (SfMUtil.cpp:39 passes 0.7)")
    ap.add_argument("--ransac-threshold", type=float, default=1.0)
    ap.add_argument("--out")
    a = ap.parse_args()
    pkg = importlib.import_module("sfm-gms_amd")
    io = importlib.import_module("sfm-gms_amd.io")
    pipeline = importlib.import_module("sfm-gms_amd.pipeline")
    ds = io.load_c(a.path)
    with pkg.GmsContext(0) as ctx:
        r = pipeline.run_dataset(ctx, ds, a.rot, a.scale, a.thr, match=True if a.match else None, camera=a.camera, dist=a.dist,
                                 prob=a.prob, ransac_threshold=a.ransac_threshold)
    res = r["results"]
    line = {"file": a.path, "frames": len(ds.frames), "pairs": len(res), "matches": int(r["pairs"]["m"].sum()),
            "kept": int(res["n_inliers"][res["status"] == 0].sum()), "failed_pairs": int((res["status"] != 0).sum()),
            "flags": [a.rot, a.scale, a.thr]}
    if "two_view" in r:
        tv = r["two_view"]
        ok = tv["status"] == 0
        fin = np.maximum(tv["n_finite"][ok], 1)
        line.update(two_view_ok=int(ok.sum()), ransac_inliers=int(tv["n_ransac"][ok].sum()), pose_inliers=int(tv["n_pose"][ok].sum()),
                    triangulated=int(tv["n_triangulated"][ok].sum()),
                    reprojection_rms=[float(np.sqrt((tv["sum_sq_err1"][ok] / fin).mean())) if ok.any() else None,
                                      float(np.sqrt((tv["sum_sq_err2"][ok] / fin).mean())) if ok.any() else None])
    if a.out:
        np.savez(a.out, **r)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
