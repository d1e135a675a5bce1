#!/usr/bin/env python3
"""Diagnostic: gms_find_essential_batch_device on small scenes, alone and in a batch, against the numpy restatement."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sfm_ref  # noqa: E402


def run(ctx, pkg, types, scenes, camera, prob):
    import torch
    from test_gpu_twoview import _coords_batch
    pairs, tv, c1, c2 = _coords_batch(pkg, scenes)
    dev = torch.device("cuda", 0)
    d_pairs = torch.from_numpy(pairs.view(np.uint8).reshape(-1)).to(dev)
    d_c1, d_c2 = torch.from_numpy(c1.reshape(-1).copy()).to(dev), torch.from_numpy(c2.reshape(-1).copy()).to(dev)
    d_tv = torch.from_numpy(tv.view(np.uint8).reshape(-1).copy()).to(dev)
    d_mask = torch.full((len(c1),), 77, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.find_essential_batch_device(types.make_camera(camera), d_pairs.data_ptr(), len(pairs), d_c1.data_ptr(), d_c2.data_ptr(), d_mask.data_ptr(),
                                    d_tv.data_ptr(), prob, 1.0, 1000)
    ctx.synchronize()
    return d_tv.cpu().numpy().view(types.TWO_VIEW_DTYPE)


def main():
    pkg = importlib.import_module("sfm-gms_amd")
    types = importlib.import_module("sfm-gms_amd.types")
    from test_gpu_twoview import _scene
    ctx = pkg.GmsContext(0)
    camera = (1400.0, 1380.0, 960.0, 540.0)
    cases = [(1, 800, 0.3), (2, 3, 0.0), (3, 3000, 0.1), (4, 60, 0.5), (5, 6, 0.0), (6, 5, 0.0), (7, 400, 0.8), (8, 0, 0.0), (9, 1500, 0.45)]
    scenes = [_scene(s_, n_, o_)[:2] for s_, n_, o_ in cases]
    for label, sub in (("batch", list(range(9))), ("alone", [4]), ("pair45", [4, 5]), ("pair34", [3, 4]), ("first5", [0, 1, 2, 3, 4])):
        got = run(ctx, pkg, types, [scenes[i] for i in sub], camera, 0.7)
        for j, i in enumerate(sub):
            E, mask, it = sfm_ref.find_essential_mat(*scenes[i], camera, 0.7, 1.0)
            print(label, "case", i, "n", len(scenes[i][0]), "diff", (np.abs(got["E"][j] - E).max() if E is not None else None), "iters", int(got["ransac_iters"][j]), it,
                  "count", int(got["n_ransac"][j]), int(mask.sum()), "status", int(got["status"][j]))
    for n in (6,):
        bad = 0
        scenes = [_scene(100 + s, n, 0.0)[:2] for s in range(40)]
        got = run(ctx, pkg, types, scenes, camera, 0.7)
        for i, (u1, u2) in enumerate(scenes):
            tr = []
            E, mask, it = sfm_ref.find_essential_mat(u1, u2, camera, 0.7, 1.0, trace=tr)
            d = np.abs(got["E"][i] - E).max() if E is not None else -1
            if d > 1e-9 or int(got["ransac_iters"][i]) != it or int(got["n_ransac"][i]) != int(mask.sum()):
                bad += 1
                if bad <= 3:
                    x1 = np.stack([(u1[:, 0].astype(np.float64) - camera[2]) / camera[0], (u1[:, 1].astype(np.float64) - camera[3]) / camera[1]], axis=1)
                    x2 = np.stack([(u2[:, 0].astype(np.float64) - camera[2]) / camera[0], (u2[:, 1].astype(np.float64) - camera[3]) / camera[1]], axis=1)
                    thr = 1.0 / ((camera[0] + camera[1]) / 2)
                    t32 = np.float32(thr * thr)
                    print(" n", n, "scene", i, "diff", d, "iters", int(got["ransac_iters"][i]), it, "count", int(got["n_ransac"][i]), int(mask.sum()))
                    for itn, idx, models in tr[:3]:
                        errs = [sfm_ref.sampson_errors(m, x1, x2) for m in models]
                        print("   iter", itn, idx, "counts", [int((e <= t32).sum()) for e in errs], "which is gpu's",
                              ["%.0e" % np.abs(m - got["E"][i]).max() for m in models])
                        for e in errs:
                            print("      errs/t", ["%.3g" % (v / t32) for v in e])
                    # the same scene alone
                    alone = run(ctx, pkg, types, [scenes[i]], camera, 0.7)
                    print("   alone: diff", np.abs(alone["E"][0] - E).max())
        print("n", n, "bad", bad, "of", len(scenes))
    ctx.close()


if __name__ == "__main__":
    main()
