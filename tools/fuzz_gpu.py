#!/usr/bin/env python3
"""Extended randomised parity sweep on the GPU (diagnostic; the committed suite is tests/test_gpu_fuzz.py): pairs of every
size class (register + LDS kernels, 16-match-per-thread variant, band / tile kernels), clustered and lattice keypoints, all
flag combinations and a few threshold factors, each compared bit for bit with the oracle.
python tools/fuzz_gpu.py [seconds] [batch]   (batch: random ragged batches through the device-resident path)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("sfm-gms_amd")
synth = importlib.import_module("sfm-gms_amd.synth")
import gms_oracle  # noqa: E402

FLAGS = [(False, False), (True, False), (False, True), (True, True)]


def case(seed):
    rng = np.random.default_rng(77000 + seed)
    w1, h1 = int(rng.integers(200, 4000)), int(rng.integers(200, 3000))
    w2, h2 = (w1, h1) if rng.uniform() < 0.6 else (int(rng.integers(200, 4000)), int(rng.integers(200, 3000)))
    size_class = int(rng.integers(0, 6))
    m = int([rng.integers(1, 4097), rng.integers(4097, 10241), rng.integers(10241, 16385), rng.integers(16385, 40000),
             rng.integers(40000, 90000), rng.integers(1, 3000)][size_class])
    n1 = int(rng.integers(max(1, m // 4), 2 * m + 2))
    n2 = int(rng.integers(max(1, m // 4), 2 * m + 2))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        xy1 = np.stack([rng.uniform(0, w1, n1), rng.uniform(0, h1, n1)], axis=1)
    elif kind == 1:  # clusters: crowded cells (the byte matrix's 255 limit, the band kernels' 65 535)
        k = int(rng.integers(1, 8))
        c = np.stack([rng.uniform(0, w1, k), rng.uniform(0, h1, k)], axis=1)
        xy1 = c[rng.integers(0, k, n1)] + rng.normal(0, min(w1, h1) / rng.uniform(8, 80), (n1, 2))
    elif kind == 2:  # lattice
        xy1 = np.stack([rng.integers(0, 41, n1) * w1 / 40.0, rng.integers(0, 41, n1) * h1 / 40.0], axis=1)
    else:  # central region
        xy1 = np.stack([rng.uniform(0.2 * w1, 0.8 * w1, n1), rng.uniform(0.1 * h1, 0.9 * h1, n1)], axis=1)
    lim1 = [np.nextafter(np.float32(w1), np.float32(0)), np.nextafter(np.float32(h1), np.float32(0))]
    lim2 = [np.nextafter(np.float32(w2), np.float32(0)), np.nextafter(np.float32(h2), np.float32(0))]
    xy1 = np.clip(xy1, 0, lim1)
    xy2 = np.stack([rng.uniform(0, w2, n2), rng.uniform(0, h2, n2)], axis=1)
    q, t = rng.integers(0, n1, m), rng.integers(0, n2, m)
    true = rng.uniform(size=m) < rng.uniform(0.1, 0.9)
    th, sc = rng.choice([0, 45, 90, 135, 180, 225, 270, 315]) * np.pi / 180, rng.choice([1.0, 0.5, 0.7071, 1.4142, 2.0])
    p = (xy1[q[true]] / [w1, h1] - 0.5)
    rot = np.stack([p[:, 0] * np.cos(th) - p[:, 1] * np.sin(th), p[:, 0] * np.sin(th) + p[:, 1] * np.cos(th)], axis=1) * (1.0 / sc if sc >= 1 else sc)
    xy2[t[true]] = (rot + 0.5) * [w2, h2] + rng.normal(0, 2, (int(true.sum()), 2))
    xy2 = np.clip(xy2, 0, lim2)
    xy1, xy2 = xy1.astype(np.float32), xy2.astype(np.float32)
    for a, lim in ((xy1, lim1), (xy2, lim2)):
        a[:, 0] = np.minimum(a[:, 0], lim[0])
        a[:, 1] = np.minimum(a[:, 1], lim[1])
    thr = float(rng.choice([6.0, 6.0, 6.0, 3.0, 1.0, 12.5]))
    return dict(size1=(w1, h1), size2=(w2, h2), kp1=synth.make_keypoints(xy1), kp2=synth.make_keypoints(xy2),
                matches=synth.make_matches(q, t, rng)), FLAGS[int(rng.integers(0, 4))], thr


def batch_main(budget):
    """Random batches through the device-resident path (one launch per batch, workspaces reused from batch to batch)."""
    batch = importlib.import_module("sfm-gms_amd.batch")
    ctx = pkg.GmsContext(0)
    t0, n, bad = time.time(), 0, 0
    seed = int(os.environ.get("FUZZ_SEED0", "0"))
    while time.time() - t0 < budget:
        rng = np.random.default_rng(123000 + seed)
        n_pairs = int(rng.integers(1, 9))
        frames, sizes, pairs, matches, off = [], [], [], [], 0
        for i in range(n_pairs):
            c, _, _ = case(seed * 16 + i)
            mt = c["matches"]
            if rng.uniform() < 0.15:
                mt = mt[:0]
            frames += [c["kp1"], c["kp2"]]
            sizes += [c["size1"], c["size2"]]
            pairs.append((2 * i, 2 * i + 1, len(mt), 0, off))
            matches.append(mt)
            off += len(mt)
        pairs = np.array(pairs, dtype=pkg.PAIR_DTYPE)
        matches = np.concatenate(matches) if off else np.zeros(0, dtype=pkg.DMATCH_DTYPE)
        flags = FLAGS[int(rng.integers(0, 4))]
        thr = float(rng.choice([6.0, 6.0, 2.0]))
        want_mask = bool(rng.integers(0, 2))
        table = batch.FrameTable(ctx, frames, sizes)
        out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, flags[0], flags[1], thr, want_mask=want_mask)
        wh = np.array(sizes, dtype=np.int32).reshape(-1)
        failed, wout, wres, wmask = gms_oracle.batch(np.concatenate(frames), table.frame_off_host, wh, pairs, matches, flags[0], flags[1], thr, 8)
        ok = failed == 0 and res.tobytes() == wres.tobytes() and (not want_mask or np.array_equal(mask, wmask))
        for i in range(len(pairs)):
            o, k = int(pairs["match_off"][i]), int(res["n_inliers"][i])
            ok = ok and out[o:o + k].tobytes() == wout[o:o + k].tobytes()
        if not ok:
            bad += 1
            print("BATCH MISMATCH seed", seed, "m", pairs["m"].tolist(), "flags", flags, "thr", thr, "mask", want_mask, res.tolist(), wres.tolist(), flush=True)
        n += 1
        seed += 1
        if n % 20 == 0:
            print(f"{n} batches, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {n} batches, {bad} mismatches")
    sys.exit(1 if bad else 0)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    if len(sys.argv) > 2 and sys.argv[2] == "batch":
        batch_main(budget)
    ctx = pkg.GmsContext(0)
    t0, n, bad = time.time(), 0, 0
    seed = int(os.environ.get("FUZZ_SEED0", "0"))
    while time.time() - t0 < budget:
        c, flags, thr = case(seed)
        got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags, thr, return_result=True)
        rc, want, _, wres = gms_oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags, thr)
        ok = rc == 0 and got.tobytes() == want.tobytes() and tuple(res)[:3] == tuple(wres)[:3]
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "m", len(c["matches"]), "flags", flags, "thr", thr, "got", len(got), tuple(res), "want", len(want), tuple(wres), flush=True)
        n += 1
        seed += 1
        if n % 50 == 0:
            print(f"{n} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {n} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
