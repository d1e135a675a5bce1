#!/usr/bin/env python3
"""Diagnostic: the GPU five-point solver (gms_selftest_five_point) against the numpy restatement on random samples, and the RANSAC
trajectory of one small scene sample by sample."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sfm_ref  # noqa: E402


def main():
    pkg = importlib.import_module("sfm-gms_amd")
    from test_twoview_core import _rot
    from test_gpu_twoview import _scene
    ctx = pkg.GmsContext(0)
    rng = np.random.default_rng(2)
    X1, X2 = [], []
    for trial in range(1500):
        R, t = _rot(rng.uniform(-0.3, 0.3, 3)), rng.uniform(-1, 1, 3)
        X = np.stack([rng.uniform(-2, 2, 5), rng.uniform(-1.5, 1.5, 5), rng.uniform(3, 9, 5)], axis=1)
        x1, Xc = X[:, :2] / X[:, 2:3], X @ R.T + t
        x2 = Xc[:, :2] / Xc[:, 2:3]
        if trial % 3 == 0:
            x2 = x2 + rng.normal(0, 0.01, (5, 2))
        if trial % 7 == 0:
            x2 = rng.uniform(-0.5, 0.5, (5, 2))
        X1.append(x1)
        X2.append(x2)
    got = ctx.selftest_five_point(np.array(X1), np.array(X2))
    off, diffs = 0, []
    for i, (x1, x2) in enumerate(zip(X1, X2)):
        want = sfm_ref.five_point(x1, x2)
        if len(want) != len(got[i]) or any(np.abs(g - w).max() > 1e-9 for g, w in zip(got[i], want)):
            off += 1
            if off <= 10:
                print("sample", i, "gpu", len(got[i]), "oracle", len(want), [float(np.abs(g - w).max()) for g, w in zip(got[i], want)])
            continue
        diffs += [np.abs(g - w).max() for g, w in zip(got[i], want)]
    print("random samples: off", off, "of", len(X1), "median", np.median(diffs), "p99", np.quantile(diffs, 0.99))
    # the scene of test_find_essential_batch case 4: n = 6
    camera = (1400.0, 1380.0, 960.0, 540.0)
    for seed, n in ((5, 6), (4, 60)):
        u1, u2 = _scene(seed, n, 0.0 if n == 6 else 0.5)[:2]
        x1 = np.stack([(u1[:, 0].astype(np.float64) - camera[2]) / camera[0], (u1[:, 1].astype(np.float64) - camera[3]) / camera[1]], axis=1)
        x2 = np.stack([(u2[:, 0].astype(np.float64) - camera[2]) / camera[0], (u2[:, 1].astype(np.float64) - camera[3]) / camera[1]], axis=1)
        tr = []
        E, mask, it = sfm_ref.find_essential_mat(u1, u2, camera, 0.7, 1.0, trace=tr)
        thr = 1.0 / ((camera[0] + camera[1]) / 2)
        t32 = np.float32(thr * thr)
        print("scene", seed, n, "oracle iters", it, "inliers", int(mask.sum()))
        for itn, idx, models in tr:
            g = ctx.selftest_five_point(x1[idx][None], x2[idx][None])[0]
            cw = [int((sfm_ref.sampson_errors(m, x1, x2) <= t32).sum()) for m in models]
            cg = [int((sfm_ref.sampson_errors(m, x1, x2) <= t32).sum()) for m in g]
            d = [float(np.abs(a - b).max()) for a, b in zip(g, models)]
            print("  iter", itn, idx, "oracle models", len(models), cw, "gpu", len(g), cg, "diffs", ["%.1e" % v for v in d])
    ctx.close()


if __name__ == "__main__":
    main()
