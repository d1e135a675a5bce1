"""-m gpu: randomised parity sweep -- many small pairs with awkward shapes (clustered keypoints, coordinates on the
grid lattice, tiny and lopsided images, few or many keypoints per match, repeated indices) through the one-shot C ABI
and, as one ragged batch, through the device-resident path; every result bit-exact against the oracle."""
import importlib

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    w1, h1 = int(rng.integers(40, 2500)), int(rng.integers(40, 2500))
    w2, h2 = (w1, h1) if rng.uniform() < 0.5 else (int(rng.integers(40, 2500)), int(rng.integers(40, 2500)))
    n1, n2 = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
    m = int(rng.integers(0, 2500))
    kind = int(rng.integers(0, 4))
    if kind == 0:    # uniform
        xy1 = np.stack([rng.uniform(0, w1, n1), rng.uniform(0, h1, n1)], axis=1)
    elif kind == 1:  # a few tight clusters
        c = np.stack([rng.uniform(0, w1, 5), rng.uniform(0, h1, 5)], axis=1)
        xy1 = c[rng.integers(0, 5, n1)] + rng.normal(0, min(w1, h1) / 60.0, (n1, 2))
    elif kind == 2:  # on the cell / half-cell lattice
        xy1 = np.stack([rng.integers(0, 41, n1) * w1 / 40.0, rng.integers(0, 41, n1) * h1 / 40.0], axis=1)
    else:            # one image row / column
        xy1 = np.stack([rng.uniform(0, w1, n1), np.full(n1, rng.uniform(0, h1))], axis=1)
    xy1 = np.clip(xy1, 0, [np.nextafter(np.float32(w1), np.float32(0)), np.nextafter(np.float32(h1), np.float32(0))])
    shift = rng.normal(0, 3, 2)
    xy2 = np.stack([rng.uniform(0, w2, n2), rng.uniform(0, h2, n2)], axis=1)
    q = rng.integers(0, n1, m)
    t = rng.integers(0, n2, m)
    # true correspondences for a part of the matches: copy the (scaled) left point into the matched right slot
    true = rng.uniform(size=m) < rng.uniform(0.2, 0.9)
    xy2[t[true]] = xy1[q[true]] * [w2 / w1, h2 / h1] + shift
    xy2 = np.clip(xy2, 0, [np.nextafter(np.float32(w2), np.float32(0)), np.nextafter(np.float32(h2), np.float32(0))])
    xy1, xy2 = xy1.astype(np.float32), xy2.astype(np.float32)
    xy1[:, 0] = np.minimum(xy1[:, 0], np.nextafter(np.float32(w1), np.float32(0)))
    xy1[:, 1] = np.minimum(xy1[:, 1], np.nextafter(np.float32(h1), np.float32(0)))
    xy2[:, 0] = np.minimum(xy2[:, 0], np.nextafter(np.float32(w2), np.float32(0)))
    xy2[:, 1] = np.minimum(xy2[:, 1], np.nextafter(np.float32(h2), np.float32(0)))
    synth = importlib.import_module("sfm-gms_amd.synth")
    return dict(size1=(w1, h1), size2=(w2, h2), kp1=synth.make_keypoints(xy1), kp2=synth.make_keypoints(xy2),
                matches=synth.make_matches(q, t, rng)), float(rng.choice([6.0, 6.0, 3.0, 0.5, 10.0]))


@pytest.mark.parametrize("block", range(8))
def test_fuzz_one_shot(ctx, oracle, block):
    for seed in range(block * 25, block * 25 + 25):
        c, thr = _random_case(seed)
        flags = cases.FLAGS[seed % 4]
        got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags, thr, return_result=True)
        rc, want, _, wres = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], *flags, thr)
        assert rc == 0, seed
        assert got.tobytes() == want.tobytes(), (seed, flags, len(got), len(want))
        assert tuple(res)[:3] == tuple(wres)[:3], seed


@pytest.mark.parametrize("rot,scale", [(False, False), (True, True)])
def test_fuzz_ragged_batch(ctx, oracle, pkg, rot, scale):
    batch = importlib.import_module("sfm-gms_amd.batch")
    frames, sizes, pairs, matches, off = [], [], [], [], 0
    for i in range(60):
        c, _ = _random_case(500 + i)
        frames += [c["kp1"], c["kp2"]]
        sizes += [c["size1"], c["size2"]]
        pairs.append((2 * i, 2 * i + 1, len(c["matches"]), 0, off))
        matches.append(c["matches"])
        off += len(c["matches"])
    pairs = np.array(pairs, dtype=pkg.PAIR_DTYPE)
    matches = np.concatenate(matches)
    table = batch.FrameTable(ctx, frames, sizes)
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, rot, scale, 6.0)
    wh = np.array(sizes, dtype=np.int32).reshape(-1)
    failed, wout, wres, wmask = oracle.batch(np.concatenate(frames), table.frame_off_host, wh, pairs, matches, rot, scale, 6.0, 4)
    assert failed == 0 and np.array_equal(mask, wmask) and res.tobytes() == wres.tobytes()
    for i in range(len(pairs)):
        o, k = int(pairs["match_off"][i]), int(res["n_inliers"][i])
        assert out[o:o + k].tobytes() == wout[o:o + k].tobytes()
