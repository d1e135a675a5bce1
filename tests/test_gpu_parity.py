"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle -- bit-exact (integer, byte and
index work: no tolerance anywhere; the fp32 normalisation and the fp64 threshold are compared bit for bit
as well)."""
import importlib

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

ADV = cases.adversarial_cases()


def _check(ctx, oracle, c, rot, scale, thr=6.0):
    got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, thr, return_result=True)
    rc, want, _, wres = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, thr)
    assert rc == 0
    assert len(got) == len(want), (len(got), len(want), res, wres)
    assert got.tobytes() == want.tobytes()
    assert (res["n_inliers"], res["best_scale"], res["best_rot"]) == \
        (wres["n_inliers"], wres["best_scale"], wres["best_rot"])
    return len(got)


# ---- BASELINE configs 1 and 2 ---------------------------------------------------------------------------
@pytest.mark.parametrize("rot,scale", cases.FLAGS)
def test_config1_plumbing_640x480_500(ctx, oracle, rot, scale):
    _check(ctx, oracle, cases.random_pair(11, n=500, size1=(640, 480), inlier_frac=0.6), rot, scale)


@pytest.mark.parametrize("rot,scale", cases.FLAGS)
@pytest.mark.parametrize("case", [0, 1, 2])
def test_config2_1080p_10k(ctx, oracle, rot, scale, case):
    theta, sc, p = [(0.0, 1.0, 0.5), (90.0, 0.5, 0.8), (45.0, 2 ** 0.5, 0.2)][case]
    kept = _check(ctx, oracle, cases.random_pair(100 + case, n=10000, inlier_frac=p, theta_deg=theta, scale=sc), rot, scale)
    if case == 0:
        assert kept > 1000


# ---- the edge cases the domain has ---------------------------------------------------
@pytest.mark.parametrize("name", sorted(ADV))
@pytest.mark.parametrize("rot,scale", cases.FLAGS)
def test_adversarial(ctx, oracle, name, rot, scale):
    _check(ctx, oracle, ADV[name], rot, scale)


@pytest.mark.parametrize("thr", [0.0, 1.5, 6.0, 13.0, 1e9])
def test_threshold_factors(ctx, oracle, thr):
    _check(ctx, oracle, cases.random_pair(51, n=4000, inlier_frac=0.4), True, False, thr)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1023, 1024, 1025, 4096, 4097, 10240, 10241, 16384])
def test_match_counts_across_kernel_variants(ctx, oracle, n):
    """m at the edges of the per-thread tiling (4 / 10 / 16 matches per thread) and of a 64-match chunk."""
    _check(ctx, oracle, cases.random_pair(60 + n % 17, n=n, inlier_frac=0.5), True, True)


# ---- BASELINE config 4 and the reference's dense call: pairs too large for the register + LDS kernel ------------
@pytest.mark.parametrize("rot,scale", cases.FLAGS)
def test_config4_4k_50k_features(ctx, oracle, rot, scale):
    c = cases.random_pair(104, n=50000, size1=(3840, 2160), inlier_frac=0.5, theta_deg=0.0 if not rot else 90.0,
                          scale=1.0 if not scale else 0.5)
    kept = _check(ctx, oracle, c, rot, scale)
    assert kept > 5000


@pytest.mark.parametrize("n", [16385, 20000, 65536])
def test_large_pairs_edges(ctx, oracle, n):
    _check(ctx, oracle, cases.random_pair(70 + n % 13, n=n, inlier_frac=0.4), True, True)


def test_dense_disparity_shape(ctx, oracle, synth, pkg):
    """DisparityUtil.cpp:123-149: one keypoint per pixel (450 x 375 = 168 750 matches in raster order), default flags."""
    w, h = 450, 375
    ys, xs = np.mgrid[0:h, 0:w]
    xy1 = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.float32)
    rng = np.random.default_rng(3)
    disp = 12.0 + 6.0 * np.sin(xy1[:, 1] / 40.0)
    xy2 = xy1.copy()
    train = np.arange(len(xy1))
    # the matcher's answer: most pixels find the pixel `disp` to the left, the rest something random
    tx = np.clip(np.rint(xy1[:, 0] - disp), 0, w - 1).astype(np.int64)
    train = (ys.ravel() * w + tx).astype(np.int64)
    bad = rng.uniform(size=len(train)) < 0.3
    train[bad] = rng.integers(0, len(train), int(bad.sum()))
    c = dict(size1=(w, h), size2=(w, h), kp1=synth.make_keypoints(xy1), kp2=synth.make_keypoints(xy2),
             matches=synth.make_matches(np.arange(len(xy1)), train, rng))
    kept = _check(ctx, oracle, c, False, False)
    assert kept > 50000


def test_large_pairs_in_a_batch(ctx, oracle, pkg, synth):
    batch = importlib.import_module("sfm-gms_amd.batch")
    size = (1920, 1080)
    frames, pairs, matches = _sequence_batch(pkg, synth, 4, 30000, 5, 33, ragged=True, size=size)
    table = batch.FrameTable(ctx, frames, [size] * len(frames))
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, True, False, 6.0)
    kp_all = np.concatenate(frames)
    wh = np.array([size] * len(frames), dtype=np.int32).reshape(-1)
    failed, wout, wres, wmask = oracle.batch(kp_all, table.frame_off_host, wh, pairs, matches, True, False, 6.0, 4)
    assert failed == 0 and np.array_equal(mask, wmask) and res.tobytes() == wres.tobytes()
    for i in range(len(pairs)):
        o, k = int(pairs["match_off"][i]), int(res["n_inliers"][i])
        assert out[o:o + k].tobytes() == wout[o:o + k].tobytes()


def test_capacity_error_is_loud(ctx, pkg):
    n = ctx.max_matches + 1
    c = cases.random_pair(61, n=100, inlier_frac=0.5)
    with pytest.raises(pkg.GmsError) as e:
        ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], np.zeros(n, dtype=pkg.DMATCH_DTYPE))
    assert e.value.code == -5


@pytest.mark.parametrize("name", sorted(cases.domain_error_cases()))
def test_domain_errors(ctx, pkg, name):
    c = cases.domain_error_cases()[name]
    for rot, scale in ((False, False), (True, True)):
        with pytest.raises(pkg.GmsError) as e:
            ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale)
        assert e.value.code == -2


def test_bad_arguments(ctx, pkg):
    c = cases.random_pair(62, n=100)
    with pytest.raises(pkg.GmsError) as e:
        ctx.match((0, 480), c["size2"], c["kp1"], c["kp2"], c["matches"])
    assert e.value.code == -1
    with pytest.raises(TypeError):
        ctx.match(c["size1"], c["size2"], np.zeros(4, dtype=np.float32), c["kp2"], c["matches"])


def test_module_level_drop_in(pkg, oracle):
    """matchGMS(...) with the reference's argument order and defaults (DisparityUtil.cpp:149 call shape)."""
    c = cases.random_pair(63, n=3000, inlier_frac=0.6)
    got = pkg.matchGMS(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"])
    rc, want, _, _ = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], False, False, 6.0)
    assert got.tobytes() == want.tobytes()
    got = pkg.matchGMS(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], True, True)  # FeatureMatchUtil.cpp:69
    rc, want, _, _ = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], True, True, 6.0)
    assert got.tobytes() == want.tobytes()


# ---- pieces ---------------------------------------------------------------------------------------------------
def test_threshold_fp64_matches_ieee(ctx, oracle):
    lib = oracle.load()
    rng = np.random.default_rng(5)
    T = rng.integers(0, 150001, 300000).astype(np.int32)
    n = rng.integers(1, 10, 300000).astype(np.int32)
    thr = np.sqrt(T.astype(np.float64) / n) * 6.0
    score = (np.floor(thr) + rng.integers(-1, 2, len(T))).astype(np.int32)  # right at the decision boundary
    got = ctx.selftest_threshold(T, n, score, 6.0)
    want = np.array([lib.gms_ref_threshold_rejects(int(a), int(b), int(c), 6.0) for a, b, c in
                     zip(T[:20000], n[:20000], score[:20000])], dtype=np.uint8)
    assert np.array_equal(got[:20000], want)
    assert np.array_equal(got, (thr > score).astype(np.uint8))  # numpy's fp64 div/sqrt/mul are IEEE too
    # exact ties: T = n * k^2, score = 6 k
    k = np.arange(1, 2000, dtype=np.int64)
    for nn in range(1, 10):
        Tt = (nn * k * k).astype(np.int32)
        ok = Tt > 0
        g = ctx.selftest_threshold(Tt[ok], np.full(ok.sum(), nn, dtype=np.int32), (6 * k[ok]).astype(np.int32), 6.0)
        assert not g.any()
    # other factors, including ones for which the squared form is not used (tiny / huge / zero / negative)
    for f in (0.0, -1.0, 1e-150, 0.37, 2.5, 5.999999999999999, 6.000000000000001, 1e9, 1e150):
        Tf = rng.integers(1, 100000, 50000).astype(np.int32)
        nf = rng.integers(1, 10, 50000).astype(np.int32)
        th = np.sqrt(Tf.astype(np.float64) / nf) * f
        sf = np.clip(np.floor(np.clip(th, 0, 2e6)) + rng.integers(-1, 2, len(Tf)), 0, 2 ** 21).astype(np.int32)
        assert np.array_equal(ctx.selftest_threshold(Tf, nf, sf, f), (th > sf).astype(np.uint8)), f


def test_normalize_kernel_is_ieee_fp32_divide(ctx, pkg):
    import torch
    batch = importlib.import_module("sfm-gms_amd.batch")
    synth = importlib.import_module("sfm-gms_amd.synth")
    rng = np.random.default_rng(8)
    sizes = [(1920, 1080), (641, 479), (3, 7), (3840, 2160)]
    frames = []
    for w, h in sizes:
        xy = np.stack([rng.uniform(0, w, 5000), rng.uniform(0, h, 5000)], axis=1).astype(np.float32)
        xy[:3] = [[0.0, 0.0], [-0.0, -0.0], [np.nextafter(np.float32(w), np.float32(0)), 0.5]]
        frames.append(synth.make_keypoints(xy))
    table = batch.FrameTable(ctx, frames, sizes)
    got = table.d_pts.cpu().numpy()[4: 4 + 2 * table.total].reshape(-1, 2)          # behind the table's 16-byte header
    for f, (w, h) in enumerate(sizes):
        lo, hi = table.frame_off_host[f], table.frame_off_host[f + 1]
        want = np.stack([frames[f]["x"] / np.float32(w), frames[f]["y"] / np.float32(h)], axis=1) + np.float32(0.0)
        assert got[lo:hi].tobytes() == want.astype(np.float32).tobytes()
    del torch


# ---- the device-resident batch path ---------------------------------------------------------------------------
def _sequence_batch(pkg, synth, n_frames, n_kp, n_pairs, seed, ragged=False, size=(1920, 1080)):
    frames = synth.make_sequence(seed, n_frames, size=size, n_kp=n_kp)
    pairs = np.zeros(n_pairs, dtype=pkg.PAIR_DTYPE)
    matches, off = [], 0
    rng = np.random.default_rng(seed)
    total = pkg.all_pairs_count(n_frames)
    for i in range(n_pairs):
        a, b = pkg.pair_from_index((i * 7919) % total, n_frames)
        mt = synth.sequence_matches(seed * 1000 + i, n_kp, n_kp, 0.5)
        if ragged:
            mt = mt[: int(rng.integers(0, n_kp + 1))]
        pairs[i] = (a, b, len(mt), 0, off)
        matches.append(mt)
        off += len(mt)
    return frames, pairs, (np.concatenate(matches) if off else np.zeros(0, dtype=pkg.DMATCH_DTYPE))


@pytest.mark.parametrize("rot,scale", [(False, False), (True, True)])
def test_batch_ragged_pairs(ctx, oracle, pkg, synth, rot, scale):
    batch = importlib.import_module("sfm-gms_amd.batch")
    size = (1280, 720)
    frames, pairs, matches = _sequence_batch(pkg, synth, 9, 3000, 40, 7, ragged=True, size=size)
    pairs["m"][3] = 0  # an empty pair in the middle
    table = batch.FrameTable(ctx, frames, [size] * len(frames))
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, rot, scale, 6.0)
    kp_all = np.concatenate(frames)
    wh = np.array([size] * len(frames), dtype=np.int32).reshape(-1)
    failed, wout, wres, wmask = oracle.batch(kp_all, table.frame_off_host, wh, pairs, matches, rot, scale, 6.0, 4)
    assert failed == 0 and (res["status"] == 0).all()
    assert np.array_equal(mask, wmask)
    assert res.tobytes() == wres.tobytes()
    for i in range(len(pairs)):
        o, k = int(pairs["match_off"][i]), int(res["n_inliers"][i])
        assert out[o:o + k].tobytes() == wout[o:o + k].tobytes()


def test_batch_one_bad_pair_does_not_poison_the_rest(ctx, oracle, pkg, synth):
    batch = importlib.import_module("sfm-gms_amd.batch")
    size = (640, 480)
    frames, pairs, matches = _sequence_batch(pkg, synth, 5, 800, 6, 12, size=size)
    matches = matches.copy()
    matches["trainIdx"][int(pairs["match_off"][2]) + 5] = 10 ** 6
    table = batch.FrameTable(ctx, frames, [size] * len(frames))
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches)
    assert res["status"].tolist() == [0, 0, -2, 0, 0, 0]
    assert res["n_inliers"][2] == 0 and res["n_inliers"][[0, 1, 3, 4, 5]].min() > 0


def test_full_size_properties(ctx, pkg, synth):
    """BASELINE config-3 shape (1080p, 10k matches/pair, many pairs per launch), checked through properties that
    need no oracle: survivors are a verbatim, order-preserving subsequence; the mask agrees with them;
    filtering the survivors of a pair again keeps a subset of them... and the launch is deterministic."""
    batch = importlib.import_module("sfm-gms_amd.batch")
    size = (1920, 1080)
    frames, pairs, matches = _sequence_batch(pkg, synth, 24, 10000, 192, 21, size=size)
    table = batch.FrameTable(ctx, frames, [size] * len(frames))
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches)
    out2, res2, mask2 = batch.filter_pairs(ctx, table, pairs, matches)
    assert out.tobytes() == out2.tobytes() and res.tobytes() == res2.tobytes() and np.array_equal(mask, mask2)
    assert (res["status"] == 0).all() and (res["n_inliers"] > 1000).all()
    for i in range(len(pairs)):
        o, m, k = int(pairs["match_off"][i]), int(pairs["m"][i]), int(res["n_inliers"][i])
        sel = mask[o:o + m].astype(bool)
        assert sel.sum() == k
        assert out[o:o + k].tobytes() == matches[o:o + m][sel].tobytes()
    # true correspondences (trainIdx == queryIdx) dominate what survives
    kept = out[: int(res["n_inliers"][0])]
    assert (kept["queryIdx"] == kept["trainIdx"]).mean() > 0.9


def test_normalize_kernel_against_the_reference_binary(ctx, pkg, synth):
    """The product kernel directly against the reference DLL: gms_normalize_device on the cv::KeyPoint coordinates of
    tests/golden/refdll_normalize.npz must produce the bits GMSMatcher::normalizePoints produced when it was executed out of
    opencv_xfeatures2d452.dll (tests/golden/make_refdll_vectors.py)."""
    import os
    batch = importlib.import_module("sfm-gms_amd.batch")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdll_normalize.npz"))
    n_cases = len([k for k in z.files if k.endswith("_size")])
    frames = [synth.make_keypoints(z[f"c{i}_xy"]) for i in range(n_cases)]
    sizes = [tuple(int(v) for v in z[f"c{i}_size"]) for i in range(n_cases)]
    table = batch.FrameTable(ctx, frames, sizes)
    got = table.d_pts.cpu().numpy()[4: 4 + 2 * table.total].reshape(-1, 2)
    want = np.concatenate([z[f"c{i}_normalized"] for i in range(n_cases)])
    assert got.view(np.uint32).tobytes() == want.view(np.uint32).tobytes()
