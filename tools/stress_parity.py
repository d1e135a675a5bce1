#!/usr/bin/env python3
"""One-off stress (not part of the test suite): many random pairs of realistic size -- random true scale, rotation, inlier
fraction, match count up to 16 384 -- through gms_filter_device with scale hypotheses (with and without rotation), every result
compared with the oracle. Run on the GPU box: python tools/stress_parity.py [n_cases]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("sfm-gms_amd")
synth = importlib.import_module("sfm-gms_amd.synth")
batch = importlib.import_module("sfm-gms_amd.batch")
import gms_oracle  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 240
rng = np.random.default_rng(2024)
ctx = pkg.GmsContext(0)
bad, checked, by_scale = 0, 0, {}
for chunk in range(0, n_cases, 40):
    frames, sizes, pairs, matches, off = [], [], [], [], 0
    for i in range(chunk, min(chunk + 40, n_cases)):
        n = int(rng.choice([500, 2000, 4096, 4097, 7000, 10000, 10240, 10241, 16000, 16384]))
        size = (int(rng.integers(320, 4000)), int(rng.integers(240, 3000)))
        kp1, kp2, m = synth.make_pair(7000 + i, size1=size, n1=n, inlier_frac=float(rng.uniform(0.15, 0.9)),
                                      theta_deg=float(rng.choice([0, 0, 10, 45, 90, 135, 180, 225, 270, 315])),
                                      scale=float(rng.choice([0.5, 0.7, 1.0, 1.0, 1.0, 1.4, 2.0])), noise_px=float(rng.uniform(0.3, 4.0)))
        frames += [kp1, kp2]
        sizes += [size, size]
        pairs.append((2 * (i - chunk), 2 * (i - chunk) + 1, len(m), 0, off))
        matches.append(m)
        off += len(m)
    pairs = np.array(pairs, dtype=pkg.PAIR_DTYPE)
    matches = np.concatenate(matches)
    table = batch.FrameTable(ctx, frames, sizes)
    wh = np.array(sizes, dtype=np.int32).reshape(-1)
    for rot in (False, True):
        for _ in range(2):   # twice: the second launch follows the first one's verdict
            out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, rot, True, 6.0)
        failed, wout, wres, wmask = gms_oracle.batch(np.concatenate(frames), table.frame_off_host, wh, pairs, matches, rot, True, 6.0, 16)
        ok = failed == 0 and np.array_equal(mask, wmask) and res.tobytes() == wres.tobytes()
        for i in range(len(pairs)):
            o, k = int(pairs["match_off"][i]), int(wres["n_inliers"][i])
            ok = ok and out[o:o + k].tobytes() == wout[o:o + k].tobytes()
            by_scale[int(wres["best_scale"][i])] = by_scale.get(int(wres["best_scale"][i]), 0) + 1
        checked += len(pairs)
        bad += 0 if ok else 1
print(json.dumps({"pairs_checked": checked, "chunks_with_a_mismatch": bad, "best_scale_histogram": by_scale}))
sys.exit(1 if bad else 0)
