#!/usr/bin/env python3
"""Generates tests/golden/*.npz: inputs + expected outputs of the GMS path.

The reference holds no golden vectors for this path and cannot be run here (binary-only Windows DLL,
SURVEY.md 8c), so the expected outputs come from oracle/gms_ref.c (the C restatement of the DLL) and are
only written when the independently written oracle/gms_ref_sparse.py agrees bit for bit. A fixture is
data: keypoint positions, image sizes, matches, and per flag combination the inlier mask and the winning
(scale, rotation)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases  # noqa: E402
import gms_oracle  # noqa: E402
import gms_ref_sparse  # noqa: E402


def fixture(c):
    xy1 = np.stack([c["kp1"]["x"], c["kp1"]["y"]], axis=1).astype(np.float32)
    xy2 = np.stack([c["kp2"]["x"], c["kp2"]["y"]], axis=1).astype(np.float32)
    out = dict(size1=np.array(c["size1"], dtype=np.int32), size2=np.array(c["size2"], dtype=np.int32), xy1=xy1, xy2=xy2,
               query=c["matches"]["queryIdx"].copy(), train=c["matches"]["trainIdx"].copy(),
               img_idx=c["matches"]["imgIdx"].copy(), distance=c["matches"]["distance"].copy())
    for rot, scale in cases.FLAGS:
        rc, kept, mask, res = gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0)
        assert rc == 0
        m2, s2, r2 = gms_ref_sparse.match_mask(c["size1"], c["size2"], xy1, xy2, out["query"], out["train"], rot, scale, 6.0)
        assert np.array_equal(mask, m2) and (res["best_scale"], res["best_rot"]) == (s2, r2), "restatements disagree"
        tag = f"r{int(rot)}s{int(scale)}"
        out["mask_" + tag] = np.packbits(mask)
        out["best_" + tag] = np.array([res["n_inliers"], res["best_scale"], res["best_rot"]], dtype=np.int32)
    return out


def main():
    adv = cases.adversarial_cases()
    todo = {k: adv[k] for k in ("cell_borders", "last_half_cell", "argmax_tie", "thresh_tie", "thresh_just_below",
                                "corners_edges", "permuted_duplicates", "sizes_differ", "rot45_scale_sqrt2",
                                "scale_half", "zeros_and_edges")}
    todo["config1_640x480_500"] = cases.random_pair(11, n=500, size1=(640, 480), inlier_frac=0.6)
    todo["config2_1080p_10k"] = cases.random_pair(100, n=10000, size1=(1920, 1080), inlier_frac=0.5)
    todo["config2_1080p_10k_rot90_half"] = cases.random_pair(101, n=10000, size1=(1920, 1080), inlier_frac=0.8,
                                                             theta_deg=90.0, scale=0.5)
    for name, c in todo.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fixture(c))
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    main()
