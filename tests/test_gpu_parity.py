"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle -- bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLAGS = [(False, False), (True, False), (False, True), (True, True)]


def _check(ctx, oracle, size1, size2, kp1, kp2, matches, rot, scale, thr=6.0):
    got, res = ctx.match(size1, size2, kp1, kp2, matches, rot, scale, thr, return_result=True)
    rc, want, _, wres = oracle.match(size1, size2, kp1, kp2, matches, rot, scale, thr)
    assert rc == 0
    assert len(got) == len(want), (len(got), len(want), res, wres)
    assert got.tobytes() == want.tobytes()
    assert (res["n_inliers"], res["best_scale"], res["best_rot"]) == \
        (wres["n_inliers"], wres["best_scale"], wres["best_rot"])
    return len(got)


@pytest.mark.parametrize("rot,scale", FLAGS)
def test_config1_plumbing_640x480_500(ctx, oracle, synth, rot, scale):
    size = (640, 480)
    kp1, kp2, m = synth.make_pair(11, size1=size, n1=500, inlier_frac=0.6)
    _check(ctx, oracle, size, size, kp1, kp2, m, rot, scale)


@pytest.mark.parametrize("rot,scale", FLAGS)
@pytest.mark.parametrize("case", [0, 1, 2])
def test_config2_1080p_10k(ctx, oracle, synth, rot, scale, case):
    size = (1920, 1080)
    theta, sc, p = [(0.0, 1.0, 0.5), (90.0, 0.5, 0.8), (45.0, 2 ** 0.5, 0.2)][case]
    kp1, kp2, m = synth.make_pair(100 + case, size1=size, n1=10000, inlier_frac=p, theta_deg=theta, scale=sc)
    kept = _check(ctx, oracle, size, size, kp1, kp2, m, rot, scale)
    if case == 0:
        assert kept > 1000


def test_threshold_fp64_matches_ieee(ctx, oracle):
    lib = oracle.load()
    rng = np.random.default_rng(5)
    T = rng.integers(0, 90001, 200000).astype(np.int32)
    n = rng.integers(1, 10, 200000).astype(np.int32)
    # scores right at the decision boundary
    thr = np.sqrt(T.astype(np.float64) / n) * 6.0
    score = (np.floor(thr) + rng.integers(-1, 2, len(T))).astype(np.int32)
    got = ctx.selftest_threshold(T, n, score, 6.0)
    want = np.array([lib.gms_ref_threshold_rejects(int(a), int(b), int(c), 6.0) for a, b, c in
                     zip(T[:20000], n[:20000], score[:20000])], dtype=np.uint8)
    assert np.array_equal(got[:20000], want)
    # numpy's fp64 sqrt/div/mul are IEEE as well: check the whole set
    assert np.array_equal(got, (thr > score).astype(np.uint8))
