"""-m gpu: the byte-matrix fast path of the filter (no scale hypotheses, every left cell <= 255 matches) and the
hand-over to the hashed path on the other side of each of its limits, against the CPU oracle, bit-exact. The cases
sit right at the limits: 255 / 256 matches in a cell of the unshifted grid, in a cell that only exists under a shifted
grid type, in one half cell (the byte histogram wraps at 256), one matrix entry reaching 255, a frame too large to
stage in LDS, and eligible and ineligible pairs side by side in one launch."""
import importlib

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

W, H = 2000, 1000  # 100 x 50 pixels per left cell


def _blob(rng, cx, cy, n, half_w=0.2, half_h=0.2):
    """n points around (cx, cy), in units of left cells."""
    return np.stack([(cx + rng.uniform(-half_w, half_w, n)) * W / 20.0, (cy + rng.uniform(-half_h, half_h, n)) * H / 20.0], axis=1)


def _case(groups, seed=5, background=True, shuffle=True):
    """groups: [(left (cx, cy) in cells, right (cx, cy) in cells, n, half extents)], plus an identity background of
    coherent matches (so that cells pass the threshold) and a little noise."""
    rng = np.random.default_rng(seed)
    left, right = [], []
    for (lc, rc, n, ext) in groups:
        left.append(_blob(rng, lc[0], lc[1], n, *ext))
        right.append(_blob(rng, rc[0], rc[1], n, *ext))
    if background:
        for cy in range(2, 18):
            for cx in range(2, 18):
                n = int(rng.integers(8, 24))
                d = rng.uniform(-0.45, 0.45, (n, 2))
                left.append(np.stack([(cx + 0.5 + d[:, 0]) * W / 20.0, (cy + 0.5 + d[:, 1]) * H / 20.0], axis=1))
                right.append(left[-1] + rng.uniform(-3, 3, (n, 2)))
        left.append(np.stack([rng.uniform(0, W - 1, 400), rng.uniform(0, H - 1, 400)], axis=1))
        right.append(np.stack([rng.uniform(0, W - 1, 400), rng.uniform(0, H - 1, 400)], axis=1))
    xy1 = np.clip(np.concatenate(left), 0, [W - 0.01, H - 0.01]).astype(np.float32)
    xy2 = np.clip(np.concatenate(right), 0, [W - 0.01, H - 0.01]).astype(np.float32)
    idx = np.arange(len(xy1))
    c = cases._pair(xy1, xy2, idx, idx, (W, H), (W, H))
    if shuffle:
        c["matches"] = c["matches"][rng.permutation(len(idx))]
    return c


def _check(ctx, oracle, c, rot, scale=False, thr=6.0):
    got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, thr, return_result=True)
    rc, want, _, wres = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, thr)
    assert rc == 0
    assert got.tobytes() == want.tobytes(), (len(got), len(want), res, wres)
    assert (res["n_inliers"], res["best_scale"], res["best_rot"]) == (wres["n_inliers"], wres["best_scale"], wres["best_rot"])
    return len(got)


TIGHT = (0.2, 0.2)       # stays inside one quarter of a cell when centred on x.25 / x.75
CELL = (0.45, 0.45)      # spread over the whole unshifted cell


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("n", [254, 255, 256, 257, 400])
def test_cell_population_at_the_byte_limit(ctx, oracle, n, rot):
    """One cell of the unshifted grid with n matches (the background leaves cells 0 and 1 of the grid empty)."""
    kept = _check(ctx, oracle, _case([((0.5, 0.5), (3.5, 7.5), n, CELL)], background=True), rot)
    assert kept >= n  # the crowded cell itself passes: its n coherent matches are kept


@pytest.mark.parametrize("n", [127, 128, 200])
def test_cell_that_only_a_shifted_grid_type_overfills(ctx, oracle, n):
    """Two neighbouring unshifted cells hold n matches each in their facing halves: every cell of grid type 1 stays
    at n <= 255, the grid-type-2 cell across the border holds 2n."""
    groups = [((0.75, 0.5), (5.75, 9.5), n, (0.2, 0.4)), ((1.25, 0.5), (6.25, 9.5), n, (0.2, 0.4))]
    _check(ctx, oracle, _case(groups), False)
    _check(ctx, oracle, _case(groups), True)


@pytest.mark.parametrize("n", [255, 256, 300, 513, 1030])
def test_half_cell_histogram_wrap(ctx, oracle, n):
    """n matches inside ONE half cell: the byte histogram wraps at 256 (and again at 512, 768, 1024); a wrapped pair must
    still be recognised as not representable."""
    _check(ctx, oracle, _case([((0.25, 0.25), (10.25, 10.25), n, TIGHT)]), False)


def test_one_matrix_entry_reaches_255(ctx, oracle):
    """255 matches from one left cell into one right cell: the byte of that entry ends at its maximum, and the row's
    arg-max key at (255 - 255) << 11."""
    c = _case([((0.5, 0.5), (12.5, 3.5), 255, (0.3, 0.3))], background=True)
    kept = _check(ctx, oracle, c, False)
    assert kept >= 255


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("n", [256, 257, 300, 600])
def test_one_matrix_entry_above_255(ctx, oracle, n, rot):
    """More than 255 matches from one left cell into one right cell: in the crowded mode the entry's byte would wrap; the pair
    has to be recognised and handed to the general path."""
    c = _case([((0.5, 0.5), (12.5, 3.5), n, (0.3, 0.3))], background=True)
    assert _check(ctx, oracle, c, rot) >= n


@pytest.mark.parametrize("rot", [False, True])
def test_crowded_scene_stays_on_the_byte_matrix(ctx, oracle, rot):
    """Many cells far above 255 matches but spread over many right cells each (noisy correspondences): no entry reaches 255, the
    crowded mode (16-bit nLeft counters) finishes the pair."""
    rng = np.random.default_rng(31)
    n = 9000
    xy1 = np.stack([rng.uniform(0.35 * W, 0.65 * W, n), rng.uniform(0.35 * H, 0.65 * H, n)], axis=1)   # 6 x 6 cells, 250 each
    xy2 = xy1 + rng.normal(0, 60.0, xy1.shape)                                                          # smeared over neighbouring right cells
    xy1 = np.clip(xy1, 0, [W - 0.01, H - 0.01]).astype(np.float32)
    xy2 = np.clip(xy2, 0, [W - 0.01, H - 0.01]).astype(np.float32)
    c = cases._pair(xy1, xy2, np.arange(n), np.arange(n), (W, H), (W, H))
    c["matches"] = c["matches"][rng.permutation(n)]
    _check(ctx, oracle, c, rot)


def test_all_matches_in_the_last_half_cell(ctx, oracle):
    """Points with x, y in the last half cell: binned under grid type 1 only (x >= 20 || y >= 20 elsewhere)."""
    _check(ctx, oracle, _case([((19.8, 19.8), (19.8, 19.8), 200, (0.15, 0.15))], background=False), False)
    _check(ctx, oracle, _case([((19.8, 5.5), (19.8, 5.5), 200, (0.15, 0.3)), ((5.5, 19.8), (5.5, 19.8), 200, (0.3, 0.15))]), True)


def test_frame_too_large_to_stage(ctx, oracle):
    """Frame B with more keypoints than the matrix area can stage (20 200): the train-side gather reads global memory."""
    rng = np.random.default_rng(9)
    n2, m = 20500, 6000
    xy2 = np.stack([rng.uniform(0, W - 1, n2), rng.uniform(0, H - 1, n2)], axis=1).astype(np.float32)
    train = rng.permutation(n2)[:m]
    xy1 = np.clip(xy2[train] + rng.uniform(-2, 2, (m, 2)), 0, [W - 0.01, H - 0.01]).astype(np.float32)
    xy1[m // 2:] = np.stack([rng.uniform(0, W - 1, m - m // 2), rng.uniform(0, H - 1, m - m // 2)], axis=1)
    c = cases._pair(xy1, xy2, np.arange(m), train, (W, H), (W, H))
    assert _check(ctx, oracle, c, False) > 1000


def test_frames_beyond_the_staging_area(ctx, oracle):
    """With 16-bit code words the matrix area stages 80 800 keypoints of the two frames together; a frame B of 90 000 keypoints
    (6 000 matches: still the register kernel's pair) makes both gathers read global memory -- default flags (dense_pair_plain) and
    with rotation hypotheses (dense_pair)."""
    rng = np.random.default_rng(19)
    n2, m = 90000, 6000
    xy2 = np.stack([rng.uniform(0, W - 1, n2), rng.uniform(0, H - 1, n2)], axis=1).astype(np.float32)
    train = rng.permutation(n2)[:m]
    xy1 = np.clip(xy2[train] + rng.uniform(-2, 2, (m, 2)), 0, [W - 0.01, H - 0.01]).astype(np.float32)
    xy1[m // 2:] = np.stack([rng.uniform(0, W - 1, m - m // 2), rng.uniform(0, H - 1, m - m // 2)], axis=1)
    c = cases._pair(xy1, xy2, np.arange(m), train, (W, H), (W, H))
    assert _check(ctx, oracle, c, False) > 1000
    assert _check(ctx, oracle, c, True) > 1000


def test_eligible_and_ineligible_pairs_share_a_launch(ctx, oracle):
    """A batch whose pairs alternate between the byte-matrix path and the general path (a 300-match cell)."""
    batch = importlib.import_module("sfm-gms_amd.batch")
    types = importlib.import_module("sfm-gms_amd.types")
    cs = []
    for i in range(12):
        groups = [((0.5, 0.5), (3.5 + i, 7.5), 300 if i % 3 == 1 else 40 + i, CELL)]
        cs.append(_case(groups, seed=20 + i))
    frames = batch.FrameTable(ctx, [c["kp1"] for c in cs] + [c["kp2"] for c in cs], [c["size1"] for c in cs] + [c["size2"] for c in cs])
    pairs = np.zeros(len(cs), dtype=types.PAIR_DTYPE)
    off = 0
    for i, c in enumerate(cs):
        pairs[i]["frame_a"], pairs[i]["frame_b"], pairs[i]["m"], pairs[i]["match_off"] = i, len(cs) + i, len(c["matches"]), off
        off += len(c["matches"])
    matches = np.concatenate([c["matches"] for c in cs])
    for rot in (False, True):
        out, results, mask = batch.filter_pairs(ctx, frames, pairs, matches, rot, False, 6.0)
        for i, c in enumerate(cs):
            rc, want, wmask, wres = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, False, 6.0)
            o, k = int(pairs[i]["match_off"]), int(results[i]["n_inliers"])
            assert rc == 0 and results[i]["status"] == 0 and k == wres["n_inliers"] and results[i]["best_rot"] == wres["best_rot"]
            assert out[o:o + k].tobytes() == want.tobytes()
            assert np.array_equal(mask[o:o + len(c["matches"])], np.asarray(wmask, dtype=np.uint8))


def test_integer_threshold_form_agrees_with_fp64(ctx):
    """The byte-matrix path decides sqrt(T / n) * factor > score on exact integers when factor is a small integer. The
    selftest kernel evaluates both forms and reports 2 on a disagreement; the fp64 form is checked against IEEE here.
    All of the path's operand range around the decision boundary, every exact tie included."""
    rng = np.random.default_rng(12)
    for f in (1.0, 2.0, 3.0, 6.0, 7.0, 13.0, 1023.0, 1024.0, 5.5, 0.0):
        T, n = np.meshgrid(np.arange(0, 9 * 255 + 1, dtype=np.int64), np.arange(1, 10, dtype=np.int64))
        T, n = T.ravel(), n.ravel()
        th = np.sqrt(T.astype(np.float64) / n) * f
        for d in (-1, 0, 1):
            sc = np.clip(np.floor(th) + d, 0, 9 * 255).astype(np.int32)
            got = ctx.selftest_threshold(T.astype(np.int32), n.astype(np.int32), sc, f)
            assert got.max() <= 1, ("integer and fp64 forms disagree", f, d)
            assert np.array_equal(got, (th > sc).astype(np.uint8)), (f, d)
        # exact ties T * f^2 == score^2 * n are not rejections (thresh == score)
        if f >= 1 and f == int(f):
            k = np.arange(1, 48, dtype=np.int64)
            for nn in range(1, 10):
                Tt, st = nn * k * k, int(f) * k
                ok = (Tt <= 9 * 255) & (st <= 9 * 255)
                if ok.any():
                    g = ctx.selftest_threshold(Tt[ok].astype(np.int32), np.full(ok.sum(), nn, dtype=np.int32), st[ok].astype(np.int32), f)
                    assert not g.any(), (f, nn)
    # random operands, random integer factors
    for f in rng.integers(1, 1024, 20):
        T = rng.integers(0, 9 * 255 + 1, 20000).astype(np.int32)
        n = rng.integers(1, 10, 20000).astype(np.int32)
        sc = rng.integers(0, 9 * 255 + 1, 20000).astype(np.int32)
        th = np.sqrt(T.astype(np.float64) / n) * float(f)
        assert np.array_equal(ctx.selftest_threshold(T, n, sc, float(f)), (th > sc).astype(np.uint8)), f


# ---- scale hypotheses: scales 0..2 run on the byte matrix (filter_kernel_dense_scales), 3 and 4 on the hashed path ----------
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("n", [255, 256, 400])
def test_scales_cell_population_at_the_byte_limit(ctx, oracle, n, rot):
    """With 256 or more matches in a cell the first kernel leaves an empty record and the hashed kernel does all five scales."""
    _check(ctx, oracle, _case([((0.5, 0.5), (3.5, 7.5), n, CELL)], background=True), rot, scale=True)


@pytest.mark.parametrize("scale_factor,theta", [(1.0, 0.0), (0.5, 0.0), (0.7071, 90.0), (1.4142, 45.0), (2.0, 180.0)])
def test_scales_winner_on_either_side_of_the_hand_over(ctx, oracle, scale_factor, theta):
    """True scale ratios that make each of the five scale hypotheses the winner in turn: 0..2 are decided by the byte-matrix
    kernel, 3 and 4 by the hashed kernel that resumes from its record."""
    c = cases.random_pair(300 + int(scale_factor * 10), n=8000, inlier_frac=0.6, theta_deg=theta, scale=scale_factor)
    for rot in (False, True):
        got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, True, 6.0, return_result=True)
        rc, want, _, wres = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, True, 6.0)
        assert rc == 0 and got.tobytes() == want.tobytes()
        assert (res["n_inliers"], res["best_scale"], res["best_rot"]) == (wres["n_inliers"], wres["best_scale"], wres["best_rot"])


def test_scales_shifted_grid_overfill_and_half_cell_wrap(ctx, oracle):
    groups = [((0.75, 0.5), (5.75, 9.5), 200, (0.2, 0.4)), ((1.25, 0.5), (6.25, 9.5), 200, (0.2, 0.4))]
    _check(ctx, oracle, _case(groups), True, scale=True)
    _check(ctx, oracle, _case([((0.25, 0.25), (10.25, 10.25), 300, TIGHT)]), False, scale=True)


def test_scales_no_inliers_in_the_first_three_scales(ctx, oracle):
    """Random matches only: every count is small; whatever the winner, record and resumption must agree with the oracle."""
    c = cases.random_pair(77, n=6000, inlier_frac=0.0)
    for rot in (False, True):
        _check(ctx, oracle, c, rot, scale=True)


# ---- the dealt lane mapping of the byte-matrix kernel (inputs in spatial order) ------------------------------------------------------
def test_dealt_mapping_whole_file_again():
    """GMS_DEAL=1 (read once per process) forces the instantiation that deals the matches to the lanes in 8-match units: everything
    in this file and the golden fixtures must come out the same."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_dense_path.py"),
                          os.path.join(root, "tests", "test_golden.py"), os.path.join(root, "tests", "test_gpu_fuzz.py"), "-m", "gpu", "-x", "-q",
                          "-k", "not whole_file_again"], capture_output=True, text=True, timeout=1200, env=dict(os.environ, GMS_DEAL="1"))
    assert res.returncode == 0, res.stdout[-3000:]


@pytest.mark.parametrize("setting", ["-1", "0", "2,7"])
def test_touch_ahead_setting_whole_file_again(setting):
    """GMS_PREFETCH (read once per process): the byte-matrix kernel touches the match records of a later pair ahead of time -- never
    (-1), before the first grid type, or before the third one for the pair seven places on instead of one CU count on. Speed only:
    everything in this file, the golden fixtures and the API tests (batches of more pairs than CUs, ragged sizes, slices) come out
    the same."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "tests", f) for f in ("test_gpu_dense_path.py", "test_golden.py", "test_gpu_api.py")]
    res = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-x", "-q", "-k", "not again"], capture_output=True,
                         text=True, timeout=1500, env=dict(os.environ, GMS_PREFETCH=setting))
    assert res.returncode == 0, res.stdout[-3000:]


def test_touch_ahead_reaches_over_ragged_pairs(ctx, pkg, oracle, synth):
    """600 pairs (more than twice the CU count) whose sizes run from 1 match to the kernel's 10 240 in no order, some empty, the last
    one ending exactly at the end of the match array: every workgroup touches the array of the pair 256 places on (another size,
    another offset, other frames) -- the filtered bytes, results and masks are the oracle's for every pair."""
    import importlib
    batch = importlib.import_module("sfm-gms_amd.batch")
    rng = np.random.default_rng(4242)
    size, n_frames, n_kp = (1280, 720), 12, 10240
    frames = synth.make_sequence(4242, n_frames, size=size, n_kp=n_kp)
    table = batch.FrameTable(ctx, frames, [size] * n_frames)
    sizes = [int(x) for x in rng.integers(1, 10241, 600)]
    for i in (0, 5, 299, 300, 557):
        sizes[i] = 0
    sizes[1], sizes[2], sizes[-1] = 10240, 1, 10240
    pairs = np.zeros(len(sizes), dtype=pkg.PAIR_DTYPE)
    chunks, off = [], 0
    for i, m in enumerate(sizes):
        a, b = int(rng.integers(0, n_frames)), int(rng.integers(0, n_frames))
        pairs[i] = (a, b, m, 0, off)
        if m:
            chunks.append(synth.sequence_matches(90000 + i, n_kp, n_kp, 0.5)[rng.permutation(n_kp)[:m]])
        off += m
    matches = np.concatenate(chunks)
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches)
    wh = np.array([size] * n_frames, dtype=np.int32).reshape(-1)
    failed, wout, wres, wmask = oracle.batch(np.concatenate(frames), table.frame_off_host, wh, pairs, matches, False, False, 6.0, 8)
    assert failed == 0
    assert res.tobytes() == wres.tobytes()
    assert mask.tobytes() == wmask.tobytes()
    for i in range(len(pairs)):
        o, k = int(pairs["match_off"][i]), int(wres["n_inliers"][i])
        assert out[o:o + k].tobytes() == wout[o:o + k].tobytes(), i


def test_probe_switches_to_dealing_on_ordered_input(pkg, oracle, synth):
    """Cell-sorted keypoints: the first launch of a fresh context runs in list order and is followed by the probe; later launches deal.
    Same bytes either way."""
    import importlib
    batch = importlib.import_module("sfm-gms_amd.batch")
    d = importlib.import_module("sfm-gms_amd.dist")
    size, n_frames, n_kp = (1920, 1080), 6, 10000
    frames = synth.make_sequence(77, n_frames, size=size, n_kp=n_kp, spatial_order=True)
    pairs = d.pair_table(n_frames, 0, 15, n_kp)
    pairs["m"][[3, 7]] = [9999, 5000]
    matches = np.concatenate([d.synth_matches_host(k, n_kp, 0.5) for k in range(15)])
    kp_all = np.concatenate(frames)
    wh = np.array([size] * n_frames, dtype=np.int32).reshape(-1)
    with pkg.GmsContext(0) as c2:
        table = batch.FrameTable(c2, frames, [size] * n_frames)
        failed, wout, wres, wmask = oracle.batch(kp_all, table.frame_off_host, wh, pairs, matches, False, False, 6.0, 8)
        assert failed == 0
        for _ in range(3):
            out, res, mask = batch.filter_pairs(c2, table, pairs, matches)
            assert res.tobytes() == wres.tobytes() and np.array_equal(mask, wmask)
            for i in range(len(pairs)):
                o, k = int(pairs["match_off"][i]), int(res["n_inliers"][i])
                assert out[o:o + k].tobytes() == wout[o:o + k].tobytes()


# ---- the scale probe (an upper bound of a scale's inlier count that lets the scale skip) ---------------------------------------------
@pytest.mark.parametrize("setting", ["GMS_SCALE_PROBE=0", "GMS_SCALE_PROBE=1", "GMS_DENSE=0"])
def test_scale_probe_setting_whole_file_again(setting):
    """GMS_SCALE_PROBE=0 / 1 (read once per process): never probe / always probe the scale hypotheses; GMS_DENSE=0: every pair on the
    hashed kernel (which probes and evaluates in the reference's order). None of it may change a byte: this file, the parity sweep,
    the golden fixtures and the fuzz cases again."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "tests", f) for f in ("test_gpu_dense_path.py", "test_gpu_parity.py", "test_golden.py", "test_gpu_fuzz.py")]
    res = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-x", "-q", "-k", "not again"], capture_output=True,
                         text=True, timeout=1500, env=dict(os.environ, **dict([setting.split("=")])))
    assert res.returncode == 0, res.stdout[-3000:]


def test_scale_probe_verdict_changes_between_launches(pkg, oracle, synth):
    """A fresh context probes, measures how often the probe lets a scale skip and follows that verdict until it measures again
    (every sixteenth launch). First 20 launches on pairs whose best hypothesis is one of the probed scales (the right image at half
    size: the 28 x 28 grid wins, its probe cannot let it skip), then 20 on pairs at scale 1.0 (the probes of scales 3 and 4 do): every
    launch bit-exact."""
    import importlib
    batch = importlib.import_module("sfm-gms_amd.batch")
    size, n = (1920, 1080), 6000
    for factor, first_id in ((0.5, 500), (1.0, 520)):
        made = [synth.make_pair(first_id + i, size1=size, n1=n, inlier_frac=0.6, scale=factor, theta_deg=0.0 if factor == 0.5 else 45.0 * i) for i in range(4)]
        frames = [kp for kp1, kp2, _ in made for kp in (kp1, kp2)]
        pairs = np.zeros(4, dtype=pkg.PAIR_DTYPE)
        for i in range(4):
            pairs[i] = (2 * i, 2 * i + 1, n, 0, i * n)
        matches = np.concatenate([m for _, _, m in made])
        kp_all = np.concatenate(frames)
        wh = np.array([size] * 8, dtype=np.int32).reshape(-1)
        if factor == 0.5:
            c2 = pkg.GmsContext(0)
        table = batch.FrameTable(c2, frames, [size] * 8)
        failed, wout, wres, wmask = oracle.batch(kp_all, table.frame_off_host, wh, pairs, matches, True, True, 6.0, 4)
        assert failed == 0
        if factor == 0.5:
            assert (wres["best_scale"] >= 2).all()
        for _ in range(20):
            out, res, mask = batch.filter_pairs(c2, table, pairs, matches, True, True)
            assert res.tobytes() == wres.tobytes() and np.array_equal(mask, wmask)
            for i in range(4):
                o, k = i * n, int(res["n_inliers"][i])
                assert out[o:o + k].tobytes() == wout[o:o + k].tobytes()
    c2.close()


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("jitter", [0.0, 0.2])
def test_scale_hypotheses_that_tie_keep_the_reference_order(ctx, oracle, rot, jitter):
    """Every match of an identity lattice is an inlier under more than one scale hypothesis: the counts tie at M, and the reference
    keeps the FIRST hypothesis with that count (scale 0). The kernel evaluates scale 1 before scale 0: the tie must go back to 0."""
    block = [y * 20 + x for y in range(4, 16) for x in range(4, 16)]
    c = cases.lattice([(lc, lc, 40) for lc in block], jitter=jitter)   # (40 per cell: scales 0 and 1 both keep all 5760; 12 would not tie)
    got, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, True, 6.0, return_result=True)
    assert len(got) == len(c["matches"]) and res["best_scale"] == 0 and res["best_rot"] == 1
    _check(ctx, oracle, c, rot, scale=True)
