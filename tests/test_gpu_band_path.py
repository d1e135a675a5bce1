"""-m gpu: large pairs (above 16 384 matches) under the reference's default flags go through the three-band LDS kernels
(gms_kernel_band.hip); against the CPU oracle, bit-exact. Covered here: sizes across the 16k-match compaction tiles, matches
concentrated on the band borders (rows 6/7, 13/14 and their half-cell neighbours), a left cell above 65 535 matches (the pair
is flagged and finished by the HBM-slab kernel), mixed batches with ragged and unaligned pairs, the optional mask, and
out-of-domain inputs."""
import importlib

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _check(ctx, oracle, c, rot=False, scale=False, thr=6.0):
    got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, thr, return_result=True)
    rc, want, _, wres = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, thr)
    assert rc == 0
    assert got.tobytes() == want.tobytes(), (len(got), len(want), res, wres)
    assert (res["n_inliers"], res["best_scale"], res["best_rot"]) == (wres["n_inliers"], wres["best_scale"], wres["best_rot"])
    return len(got)


@pytest.mark.parametrize("n", [16385, 16400, 32768, 32769, 50000, 65536, 65537, 100000, 262144])
def test_sizes_across_compaction_tiles(ctx, oracle, n):
    kept = _check(ctx, oracle, cases.random_pair(80 + n % 11, n=n, size1=(3840, 2160), inlier_frac=0.5))
    assert kept > n // 10


@pytest.mark.parametrize("thr", [0.0, 2.5, 6.0, 40.0])
def test_threshold_factors(ctx, oracle, thr):
    _check(ctx, oracle, cases.random_pair(91, n=40000, inlier_frac=0.4), thr=thr)


def _border_case(seed, n_total, rows, size=(2000, 1000)):
    """Matches whose left points crowd the given left-grid rows (in cell units, fractional), coherent displacement."""
    rng = np.random.default_rng(seed)
    w, h = size
    n_band = n_total * 2 // 3
    ys = rng.choice(rows, n_band) + rng.uniform(-0.3, 0.3, n_band)
    xs = rng.uniform(0, 20, n_band)
    xy1 = np.stack([xs * w / 20.0, ys * h / 20.0], axis=1)
    xy1 = np.concatenate([xy1, np.stack([rng.uniform(0, w, n_total - n_band), rng.uniform(0, h, n_total - n_band)], axis=1)])
    xy1 = np.clip(xy1, 0, [w - 0.01, h - 0.01]).astype(np.float32)
    xy2 = np.clip(xy1 + rng.normal(0, 2.0, xy1.shape) + [7.0, -5.0], 0, [w - 0.01, h - 0.01]).astype(np.float32)
    bad = rng.uniform(size=n_total) < 0.4
    xy2[bad] = np.stack([rng.uniform(0, w, bad.sum()), rng.uniform(0, h, bad.sum())], axis=1).astype(np.float32)
    idx = rng.permutation(n_total)
    c = cases._pair(xy1, xy2, np.arange(n_total), np.arange(n_total), size, size)
    c["matches"] = c["matches"][idx]
    return c


@pytest.mark.parametrize("rows", [(6.5, 7.0, 7.5), (13.5, 14.0, 14.5), (6.75, 7.25, 13.75, 14.25), (0.25, 19.75)])
def test_matches_on_the_band_borders(ctx, oracle, rows):
    """Left points on and around rows 7 and 14, where one band ends and the next begins -- also for the half-cell-shifted grid
    types, whose row of a point differs from the unshifted one."""
    assert _check(ctx, oracle, _border_case(5, 60000, np.array(rows))) > 5000


def test_cell_above_65535_matches_falls_through_to_the_slab_kernel(ctx, oracle):
    """70 000 of 120 000 matches in one left cell: a 16-bit entry could wrap, the pair is flagged and the HBM-slab kernel
    produces it."""
    rng = np.random.default_rng(8)
    w, h = 2000, 1000
    n_hot, n_rest = 70000, 50000
    hot1 = np.stack([rng.uniform(1000, 1099, n_hot), rng.uniform(500, 549, n_hot)], axis=1)
    rest1 = np.stack([rng.uniform(0, w - 1, n_rest), rng.uniform(0, h - 1, n_rest)], axis=1)
    xy1 = np.concatenate([hot1, rest1]).astype(np.float32)
    xy2 = np.clip(xy1 + rng.normal(0, 1.5, xy1.shape), 0, [w - 0.01, h - 0.01]).astype(np.float32)
    n = n_hot + n_rest
    c = cases._pair(xy1, xy2, np.arange(n), np.arange(n), (w, h), (w, h))
    c["matches"] = c["matches"][rng.permutation(n)]
    assert _check(ctx, oracle, c) > n_hot


def test_mixed_batch_with_mask(ctx, oracle, pkg, synth):
    """Ragged pairs (some below 16k, some above, one empty) in one launch; odd match offsets make the mask unaligned."""
    batch = importlib.import_module("sfm-gms_amd.batch")
    size = (1920, 1080)
    n_frames, n_kp = 4, 40000
    frames = synth.make_sequence(44, n_frames, size=size, n_kp=n_kp)
    lengths = [40000, 17001, 0, 333, 25000, 39999, 16385]
    pairs = np.zeros(len(lengths), dtype=pkg.PAIR_DTYPE)
    matches, off = [], 0
    total = pkg.all_pairs_count(n_frames)
    for i, ln in enumerate(lengths):
        a, b = pkg.pair_from_index((i * 5) % total, n_frames)
        mt = synth.sequence_matches(4400 + i, n_kp, n_kp, 0.5)[:ln]
        pairs[i] = (a, b, len(mt), 0, off)
        matches.append(mt)
        off += len(mt)
    matches = np.concatenate(matches)
    table = batch.FrameTable(ctx, frames, [size] * n_frames)
    kp_all = np.concatenate(frames)
    wh = np.array([size] * n_frames, dtype=np.int32).reshape(-1)
    failed, wout, wres, wmask = oracle.batch(kp_all, table.frame_off_host, wh, pairs, matches, False, False, 6.0, 4)
    assert failed == 0
    for want_mask in (True, False):
        out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, False, False, 6.0, want_mask=want_mask)
        assert res.tobytes() == wres.tobytes()
        if want_mask:
            assert np.array_equal(mask, wmask)
        for i in range(len(pairs)):
            o, k = int(pairs["match_off"][i]), int(res["n_inliers"][i])
            assert out[o:o + k].tobytes() == wout[o:o + k].tobytes()


def test_out_of_domain_input_fails_the_pair_only(ctx, oracle, pkg, synth):
    """One large pair with an out-of-range index: that pair reports GMS_ERR_DOMAIN with nothing kept, its neighbours are intact."""
    batch = importlib.import_module("sfm-gms_amd.batch")
    size = (1920, 1080)
    n_frames, n_kp = 3, 20000
    frames = synth.make_sequence(45, n_frames, size=size, n_kp=n_kp)
    pairs = np.zeros(3, dtype=pkg.PAIR_DTYPE)
    matches, off = [], 0
    for i in range(3):
        mt = synth.sequence_matches(4500 + i, n_kp, n_kp, 0.5).copy()
        if i == 1:
            mt["trainIdx"][12345] = n_kp  # one past the frame
        pairs[i] = (i % n_frames, (i + 1) % n_frames, len(mt), 0, off)
        matches.append(mt)
        off += len(mt)
    matches = np.concatenate(matches)
    table = batch.FrameTable(ctx, frames, [size] * n_frames)
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, False, False, 6.0)
    kp_all = np.concatenate(frames)
    wh = np.array([size] * n_frames, dtype=np.int32).reshape(-1)
    failed, wout, wres, wmask = oracle.batch(kp_all, table.frame_off_host, wh, pairs, matches, False, False, 6.0, 2)
    assert failed == 1 and res["status"][1] == -2 and res["n_inliers"][1] == 0
    assert res.tobytes() == wres.tobytes() and np.array_equal(mask, wmask)
    for i in (0, 2):
        o, k = int(pairs["match_off"][i]), int(res["n_inliers"][i])
        assert k > 1000 and out[o:o + k].tobytes() == wout[o:o + k].tobytes()


# ---- rotation / scale hypotheses on large pairs: tiles of left cells, three launches per scale (gms_kernel_band.hip) -------------
ROT_SCALE = [(True, False), (False, True), (True, True)]


@pytest.mark.parametrize("rot,scale", ROT_SCALE)
@pytest.mark.parametrize("n", [16385, 40000, 131072])
def test_tiles_sizes_and_flags(ctx, oracle, n, rot, scale):
    theta, sc = (90.0, 0.5) if (rot and scale) else (45.0, 1.0) if rot else (0.0, 2.0)
    c = cases.random_pair(90 + n % 7, n=n, size1=(3840, 2160), inlier_frac=0.5, theta_deg=theta, scale=sc)
    assert _check(ctx, oracle, c, rot, scale) > n // 20


@pytest.mark.parametrize("sc,theta", [(1.0, 0.0), (0.5, 180.0), (0.7071, 270.0), (1.4142, 135.0), (2.0, 315.0)])
def test_tiles_every_scale_hypothesis_wins_once(ctx, oracle, sc, theta):
    c = cases.random_pair(120 + int(sc * 10), n=30000, inlier_frac=0.6, theta_deg=theta, scale=sc)
    got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], True, True, 6.0, return_result=True)
    rc, want, _, wres = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], True, True, 6.0)
    assert rc == 0 and got.tobytes() == want.tobytes()
    assert (res["n_inliers"], res["best_scale"], res["best_rot"]) == (wres["n_inliers"], wres["best_scale"], wres["best_rot"])


@pytest.mark.parametrize("rows", [(6.5, 7.0, 7.5, 13.5, 14.0), (4.75, 5.25, 9.75, 10.25, 14.75, 15.25), (3.75, 4.25, 7.75, 8.25, 11.75, 12.25, 15.75, 16.25)])
def test_tiles_matches_on_tile_borders(ctx, oracle, rows):
    """Left points on and around the rows where the tiles of the various scales meet (7/14, 5/10/15, 4/8/12/16), all hypotheses."""
    c = _border_case(6, 45000, np.array(rows))
    assert _check(ctx, oracle, c, True, True) > 3000
    # the same along x: transpose the case
    for k in ("kp1", "kp2"):
        x = c[k]["x"].copy()
        c[k]["x"] = c[k]["y"] * (c["size1"][0] / c["size1"][1])
        c[k]["y"] = x * (c["size1"][1] / c["size1"][0])
    _check(ctx, oracle, c, True, True)


def test_tiles_cell_above_65535_falls_through(ctx, oracle):
    rng = np.random.default_rng(9)
    w, h = 2000, 1000
    n_hot, n_rest = 66000, 14000
    hot1 = np.stack([rng.uniform(1000, 1099, n_hot), rng.uniform(500, 549, n_hot)], axis=1)
    rest1 = np.stack([rng.uniform(0, w - 1, n_rest), rng.uniform(0, h - 1, n_rest)], axis=1)
    xy1 = np.concatenate([hot1, rest1]).astype(np.float32)
    xy2 = np.clip(xy1 + rng.normal(0, 1.5, xy1.shape), 0, [w - 0.01, h - 0.01]).astype(np.float32)
    n = n_hot + n_rest
    c = cases._pair(xy1, xy2, np.arange(n), np.arange(n), (w, h), (w, h))
    c["matches"] = c["matches"][rng.permutation(n)]
    assert _check(ctx, oracle, c, True, False) > n_hot


@pytest.mark.parametrize("want_mask", [True, False])
def test_tiles_mixed_batch(ctx, oracle, pkg, synth, want_mask):
    """Ragged pairs incl. an empty one, one with no inliers and one failing, rotation + scale, with and without the mask."""
    batch = importlib.import_module("sfm-gms_amd.batch")
    size = (1920, 1080)
    n_frames, n_kp = 4, 30000
    frames = synth.make_sequence(46, n_frames, size=size, n_kp=n_kp)
    lengths = [30000, 17001, 0, 333, 25000, 16385]
    pairs = np.zeros(len(lengths), dtype=pkg.PAIR_DTYPE)
    matches, off = [], 0
    total = pkg.all_pairs_count(n_frames)
    rng = np.random.default_rng(3)
    for i, ln in enumerate(lengths):
        a, b = pkg.pair_from_index((i * 5) % total, n_frames)
        mt = synth.sequence_matches(4600 + i, n_kp, n_kp, 0.0 if i == 4 else 0.5)[:ln].copy()
        if i == 1:
            mt["queryIdx"][777] = -3  # out of range: this pair fails
        pairs[i] = (a, b, len(mt), 0, off)
        matches.append(mt)
        off += len(mt)
    matches = np.concatenate(matches)
    table = batch.FrameTable(ctx, frames, [size] * n_frames)
    kp_all = np.concatenate(frames)
    wh = np.array([size] * n_frames, dtype=np.int32).reshape(-1)
    failed, wout, wres, wmask = oracle.batch(kp_all, table.frame_off_host, wh, pairs, matches, True, True, 6.0, 4)
    assert failed == 1
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, True, True, 6.0, want_mask=want_mask)
    assert res.tobytes() == wres.tobytes(), (res, wres)
    if want_mask:
        assert np.array_equal(mask, wmask)
    for i in range(len(pairs)):
        o, k = int(pairs["match_off"][i]), int(res["n_inliers"][i])
        assert out[o:o + k].tobytes() == wout[o:o + k].tobytes()


def test_tiles_workspace_left_by_a_larger_pair(ctx, oracle):
    """A pair with rotation hypotheses after a larger one on the same context: the hypothesis mask slab beyond the second pair's
    matches still holds the first pair's bits and must not be counted (m deliberately not a multiple of 16)."""
    big = cases.random_pair(131, n=90000, inlier_frac=0.7, theta_deg=45.0)
    _check(ctx, oracle, big, True, False)
    for n in (20003, 16391, 33333):
        small = cases.random_pair(132 + n % 5, n=n, inlier_frac=0.3, theta_deg=90.0, scale=0.5)
        _check(ctx, oracle, small, True, True)
        _check(ctx, oracle, small, True, False)


# ---- beyond 262 144 matches per pair (the cap of round 1): the reference's third call site, DisparityUtil.cpp:299, runs matchGMS on
#      one keypoint per pixel of a 2594 x 1131 image pair -----------------------------------------------------------------------------
def _per_pixel_case(w, h, seed, bad_frac=0.3):
    """DisparityUtil.cpp:280-299: keypoints pushed column by column (i over cols, j over rows), M = W * H matches."""
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="ij")
    xy1 = np.stack([xs.ravel(), ys.ravel()], axis=1)
    rng = np.random.default_rng(seed)
    disp = np.rint(14.0 + 9.0 * np.sin(xy1[:, 1] / 90.0)).astype(np.int64)
    tx = np.clip(xy1[:, 0].astype(np.int64) - disp, 0, w - 1)
    train = tx * h + xy1[:, 1].astype(np.int64)
    bad = rng.uniform(size=len(train)) < bad_frac
    train[bad] = rng.integers(0, len(train), int(bad.sum()))
    m = np.zeros(len(train), dtype=cases.types.DMATCH_DTYPE)
    m["queryIdx"], m["trainIdx"] = np.arange(len(train)), train
    m["distance"] = (np.arange(len(train)) % 1000).astype(np.float32)
    kp = cases.synth.make_keypoints(xy1)
    return dict(size1=(w, h), size2=(w, h), kp1=kp, kp2=kp.copy(), matches=m)


def test_portrait_mode_call_site_2594x1131_per_pixel(ctx, oracle):
    c = _per_pixel_case(2594, 1131, 4)
    assert len(c["matches"]) == 2933814
    assert _check(ctx, oracle, c) > 1500000


@pytest.mark.parametrize("rot,scale", [(True, False), (True, True)])
def test_half_a_million_matches_with_hypotheses(ctx, oracle, rot, scale):
    c = _per_pixel_case(800, 640, 5)
    assert _check(ctx, oracle, c, rot, scale) > 200000


def test_flagged_pair_beyond_the_lds_mask_limit(ctx, oracle):
    """90 000 of 400 000 matches in one left cell: the slab kernel takes the pair, its winner mask in the slab instead of LDS."""
    rng = np.random.default_rng(18)
    w, h = 2000, 1000
    n_hot, n_rest = 90000, 310000
    hot1 = np.stack([rng.uniform(1000, 1099, n_hot), rng.uniform(500, 549, n_hot)], axis=1)
    rest1 = np.stack([rng.uniform(0, w - 1, n_rest), rng.uniform(0, h - 1, n_rest)], axis=1)
    xy1 = np.concatenate([hot1, rest1]).astype(np.float32)
    xy2 = np.clip(xy1 + rng.normal(0, 1.5, xy1.shape), 0, [w - 0.01, h - 0.01]).astype(np.float32)
    n = n_hot + n_rest
    c = cases._pair(xy1, xy2, np.arange(n), np.arange(n), (w, h), (w, h))
    c["matches"] = c["matches"][rng.permutation(n)]
    for rot, scale in ((False, False), (True, True)):
        assert _check(ctx, oracle, c, rot, scale) > n_hot


def test_more_matches_in_one_cell_than_a_slot_counts(ctx, pkg):
    """2.2 M matches of one left cell: beyond the 21-bit slot counters of the slab kernel -> GMS_ERR_CAPACITY, not a wrong answer."""
    n = 2200000
    xy = np.full((4, 2), 10.0, dtype=np.float32)
    kp = cases.synth.make_keypoints(xy)
    m = np.zeros(n, dtype=pkg.DMATCH_DTYPE)
    with pytest.raises(pkg.GmsError) as e:
        ctx.match((640, 480), (640, 480), kp, kp.copy(), m)
    assert e.value.code == -5


def test_slab_kernel_alone_whole_file_again():
    """GMS_BAND=0 (read once per process): large pairs skip the LDS band / tile kernels and run on the HBM-slab kernel alone. Same
    bytes for everything in this file."""
    import os
    import subprocess
    import sys
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "not again"],
                         capture_output=True, text=True, timeout=1500, env=dict(os.environ, GMS_BAND="0"))
    assert res.returncode == 0, res.stdout[-3000:]


def test_band_and_tile_kernels_whole_file_again():
    """GMS_STREAM=0 (read once per process): pairs of 16 385 .. 65 536 matches with rotation / scale hypotheses skip the streamed
    byte-matrix kernels (their default since round 3) and run on the 16-bit tile kernels, as larger pairs always do. Same bytes for
    everything in this file."""
    import os
    import subprocess
    import sys
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "not again"],
                         capture_output=True, text=True, timeout=1500, env=dict(os.environ, GMS_STREAM="0"))
    assert res.returncode == 0, res.stdout[-3000:]


def test_streamed_default_flags_frame_a_beyond_its_staging_area(ctx, oracle):
    """stream_plain_kernel stages frame A's left codes in LDS up to 77 600 keypoints; a frame A of 90 000 keypoints with 20 000
    matches takes the left codes from global memory instead (and 20 000 matches all stay in the kernel's registers: nothing is streamed);
    the same frames with 60 000 matches stream the tail of every thread's code words as well."""
    rng = np.random.default_rng(23)
    size, n1 = (3840, 2160), 90000
    xy1 = np.stack([rng.uniform(0, size[0] - 1, n1), rng.uniform(0, size[1] - 1, n1)], axis=1).astype(np.float32)
    xy2 = np.clip(xy1 + rng.normal(0, 2.0, xy1.shape) + [9.0, -6.0], 0, [size[0] - 0.01, size[1] - 0.01]).astype(np.float32)
    for m in (20000, 60000):
        q = rng.permutation(n1)[:m]
        t = np.where(rng.uniform(size=m) < 0.5, q, rng.integers(0, n1, m))
        c = cases._pair(xy1, xy2, q, t, size, size)
        assert _check(ctx, oracle, c) > m // 10


def test_older_streamed_default_flags_kernel_whole_file_again():
    """GMS_STREAM_PLAIN=0 (read once per process): default-flags pairs of 16 385 .. 65 536 matches run on stream_dense_kernel<false>
    (round 3's first streamed kernel) instead of stream_plain_kernel. Same bytes for everything in this file."""
    import os
    import subprocess
    import sys
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "not again"],
                         capture_output=True, text=True, timeout=1500, env=dict(os.environ, GMS_STREAM_PLAIN="0"))
    assert res.returncode == 0, res.stdout[-3000:]


def test_streamed_kernels_when_an_entry_leaves_its_byte(ctx, oracle, synth):
    """40 000 matches, a third of them from one left cell to one right cell: that (left cell, right cell) entry passes 255 under every
    grid type, the streamed byte-matrix kernels flag the pair and the HBM-slab kernel produces it -- same bytes; in a batch with a pair
    the streamed kernels keep."""
    import importlib
    batch = importlib.import_module("sfm-gms_amd.batch")
    pkg = importlib.import_module("sfm-gms_amd")
    rng = np.random.default_rng(77)
    size, n = (1920, 1080), 40000
    xy1 = np.stack([rng.uniform(0, size[0] - 1, n), rng.uniform(0, size[1] - 1, n)], axis=1).astype(np.float32)
    xy2 = np.clip(xy1 + rng.normal(0, 2.0, (n, 2)).astype(np.float32) + np.float32(5.0), 0, [size[0] - 1.01, size[1] - 1.01]).astype(np.float32)
    hot = slice(0, n // 3)
    xy1[hot] = np.stack([rng.uniform(965, 1050, n // 3), rng.uniform(545, 590, n // 3)], axis=1)   # inside one 96 x 54 cell
    xy2[hot] = xy1[hot] + np.float32(3.0)
    train = np.arange(n)
    wrong = rng.uniform(size=n) < 0.3
    train[wrong] = rng.integers(0, n, int(wrong.sum()))
    kp1, kp2 = synth.make_keypoints(xy1), synth.make_keypoints(xy2)
    m_hot = synth.make_matches(np.arange(n), train, rng)
    cool1, cool2, m_cool = synth.make_pair(801, size1=size, n1=30000, inlier_frac=0.5)
    frames = [kp1, kp2, cool1, cool2]
    pairs = np.zeros(2, dtype=pkg.PAIR_DTYPE)
    pairs[0], pairs[1] = (0, 1, n, 0, 0), (2, 3, 30000, 0, n)
    matches = np.concatenate([m_hot, m_cool])
    table = batch.FrameTable(ctx, frames, [size] * 4)
    for rot, scale in ((False, False), (True, True)):
        out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, rot, scale, 6.0)
        for i in range(2):
            a, b, m, o = (int(pairs[k][i]) for k in ("frame_a", "frame_b", "m", "match_off"))
            rc, want, wmask, wres = oracle.match(size, size, frames[a], frames[b], matches[o:o + m], rot, scale, 6.0)
            assert rc == 0 and res[i].tobytes() == wres.tobytes() and np.array_equal(mask[o:o + m], wmask)
            assert out[o:o + len(want)].tobytes() == want.tobytes() and len(want) > 5000


@pytest.mark.parametrize("n", [16385, 24576, 24577, 40959, 40961, 57344, 65535, 65536])
def test_hypotheses_across_the_marking_tiles(ctx, oracle, n):
    """Rotation + scale hypotheses on the streamed kernels: the marking pass and the copy-out work on tiles of 8192 matches in the
    original order (per-tile counts of every hypothesis are the copy-out's offsets) -- sizes on and around the tile borders."""
    c = cases.random_pair(300 + n % 13, n=n, size1=(3840, 2160), inlier_frac=0.5, theta_deg=90.0 if n % 2 else 0.0, scale=0.5 if n % 3 == 0 else 1.0)
    kept = _check(ctx, oracle, c, True, True)
    assert kept > n // 8


def test_pooled_entry_above_255(ctx, oracle, synth):
    """The 10 x 10 and 14 x 14 matrices of the streamed kernels are the 20 x 20 and 28 x 28 ones' rows summed two by two, as 16-bit
    entries: 30 000 matches of which 800 go from one left cell to four neighbouring 20 x 20 right cells that are ONE 10 x 10 cell --
    every byte entry stays below 256, the pooled entry does not. Same bytes as the oracle, with the 10 x 10 hypothesis winning."""
    rng = np.random.default_rng(78)
    size, n = (1920, 1080), 30000
    xy1 = np.stack([rng.uniform(0, size[0] - 1, n), rng.uniform(0, size[1] - 1, n)], axis=1).astype(np.float32)
    # the right image at half the magnification: a left cell maps into a quarter of a 20 x 20 right cell's area ... spread on purpose
    xy2 = np.clip(xy1 * np.float32(0.5) + rng.normal(0, 1.5, (n, 2)).astype(np.float32), 0, [size[0] - 1.01, size[1] - 1.01]).astype(np.float32)
    hot = slice(0, 800)
    xy1[hot] = np.stack([rng.uniform(970, 1050, 800), rng.uniform(545, 590, 800)], axis=1)          # inside one 96 x 54 left cell
    xy2[hot] = np.stack([rng.uniform(390, 570, 800), rng.uniform(220, 320, 800)], axis=1)           # 20 x 20 cells (4..5, 4..5) = 10 x 10 cell (2, 2)
    train = np.arange(n)
    wrong = rng.uniform(size=n) < 0.3
    wrong[hot] = False
    train[wrong] = rng.integers(0, n, int(wrong.sum()))
    c = dict(size1=size, size2=size, kp1=synth.make_keypoints(xy1), kp2=synth.make_keypoints(xy2), matches=synth.make_matches(np.arange(n), train, rng))
    for rot, scale in ((False, True), (True, True)):
        got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0, return_result=True)
        rc, want, _, wres = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0)
        assert rc == 0 and got.tobytes() == want.tobytes(), (len(got), len(want), res, wres)
        assert (res["n_inliers"], res["best_scale"], res["best_rot"]) == (wres["n_inliers"], wres["best_scale"], wres["best_rot"])
        assert len(want) > 5000 and int(wres["best_scale"]) == 1  # (the 10 x 10 grid wins: its pooled entries decided the result)


def test_hypotheses_batch_with_mask_ragged_pairs(ctx, oracle, pkg, synth):
    """Five pairs of 17 000 ... 61 000 matches at unaligned offsets in one launch with rotation + scale hypotheses, the optional mask
    requested: survivors, results and mask bytes equal the oracle's, pair by pair (one of the pairs is empty of inliers by construction)."""
    batch = importlib.import_module("sfm-gms_amd.batch")
    size = (3840, 2160)
    ms = [17001, 61000, 23456, 40000, 33333]
    frames, chunks, pairs = [], [], np.zeros(len(ms), dtype=pkg.PAIR_DTYPE)
    off = 7
    for i, m in enumerate(ms):
        a, b, mm = synth.make_pair(900 + i, size1=size, n1=m, inlier_frac=0.0 if i == 2 else 0.55)
        frames += [a, b]
        pairs[i] = (2 * i, 2 * i + 1, m, 0, off)
        chunks.append((off, mm))
        off += m + 3 * (i + 1)
    matches = np.zeros(off, dtype=chunks[0][1].dtype)
    for o, mm in chunks:
        matches[o:o + len(mm)] = mm
    table = batch.FrameTable(ctx, frames, [size] * len(frames))
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, True, True, 6.0)
    for i, m in enumerate(ms):
        o = int(pairs["match_off"][i])
        rc, want, wmask, wres = oracle.match(size, size, frames[2 * i], frames[2 * i + 1], matches[o:o + m], True, True, 6.0)
        assert rc == 0 and res[i].tobytes() == wres.tobytes(), (i, res[i], wres)
        assert np.array_equal(mask[o:o + m], wmask) and out[o:o + len(want)].tobytes() == want.tobytes()
