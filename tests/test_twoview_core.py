"""CPU: the per-lane arithmetic of the two-view kernels (sfm-gms_amd/csrc/twoview_core.h: five-point solver, cv::RNG, RANSACUpdateNumIters,
the epipolar error, decomposeEssentialMat) compiled for the host by g++ (tests/cpp/twoview_host.cpp -- a test build, the product runs it
on the GPU only) against the numpy restatement oracle/sfm_ref.py, which goes about the same mathematics by other means (SVD null space,
LU solve, companion-matrix roots, SVD null vector). findEssentialMat lives in opencv_world452 (an import library in the reference):
parity unpinned; these tests pin the two implementations to each other."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import sfm_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tvh(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("tvh") / "libtvh.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror",
                           "-I" + os.path.join(ROOT, "sfm-gms_amd", "csrc"), "-o", so, os.path.join(ROOT, "tests", "cpp", "twoview_host.cpp")])
    lib = C.CDLL(so)
    vp = C.c_void_p
    lib.tvh_five_point.argtypes = [vp] * 5
    lib.tvh_rng.argtypes = [C.c_uint64, C.c_int, C.c_int, vp]
    lib.tvh_update_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    lib.tvh_decompose.argtypes = [vp] * 4
    lib.tvh_errors.argtypes = [vp, vp, vp, C.c_int, vp]
    lib.tvh_find_essential.argtypes = [vp, vp, C.c_int, vp, C.c_double, C.c_double, C.c_int, vp, vp, vp]
    return lib


def _rot(a):
    cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    return (np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
            np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]))


def _five(lib, x1, x2):
    out = np.zeros(90)
    a = [np.ascontiguousarray(v) for v in (x1[:, 0], x1[:, 1], x2[:, 0], x2[:, 1])]
    n = lib.tvh_five_point(*[v.ctypes.data for v in a], out.ctypes.data)
    return out[:9 * n].reshape(n, 3, 3)


def test_five_point_solver_against_the_restatement(tvh):
    """1500 minimal samples -- exact two-view geometry, noisy, and unrelated points: the same number of models, the same matrices (both
    sides polish every solution on the constraints themselves, so they meet at rounding level), in the same order. A sample in a few
    thousand is near-degenerate (coinciding roots) and the two root finders part ways on it: at most 0.3 % may."""
    rng = np.random.default_rng(2)
    diffs, off, exact_hit = [], 0, 0
    for trial in range(1500):
        R, t = _rot(rng.uniform(-0.3, 0.3, 3)), rng.uniform(-1, 1, 3)
        X = np.stack([rng.uniform(-2, 2, 5), rng.uniform(-1.5, 1.5, 5), rng.uniform(3, 9, 5)], axis=1)
        x1, Xc = X[:, :2] / X[:, 2:3], X @ R.T + t
        x2 = Xc[:, :2] / Xc[:, 2:3]
        if trial % 3 == 0:
            x2 = x2 + rng.normal(0, 0.01, (5, 2))
        if trial % 7 == 0:
            x2 = rng.uniform(-0.5, 0.5, (5, 2))
        want, got = sfm_ref.five_point(x1, x2), _five(tvh, x1, x2)
        if len(got) != len(want) or any(np.abs(g - w).max() > 1e-9 for g, w in zip(got, want)):
            off += 1
            continue
        diffs += [np.abs(g - w).max() for g, w in zip(got, want)]
        for g in got:                                      # every model satisfies what defines it
            h1, h2 = np.c_[x1, np.ones(5)], np.c_[x2, np.ones(5)]
            assert np.abs((h2 @ g * h1).sum(1)).max() < 1e-12 and abs(np.linalg.norm(g) - 1) < 1e-14 and np.abs(g).max() == g.reshape(-1)[np.argmax(np.abs(g))]
        if trial % 3 and trial % 7:                        # exact data: the true essential matrix is among the models
            tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
            Et = sfm_ref.canonical_sign(tx @ R / np.linalg.norm(tx @ R))
            exact_hit += any(np.abs(g - Et).max() < 1e-8 for g in got)
    assert off <= 4 and len(diffs) > 4000 and np.median(diffs) < 1e-13 and np.quantile(diffs, 0.99) < 1e-11
    assert exact_hit >= 850


def test_rng_samples_and_iteration_bound(tvh):
    """cv::RNG((uint64)-1): the first value is 2^32 - 4164903691 (one multiply-with-carry step from the all-ones state); samples of five
    distinct indices and RANSACUpdateNumIters agree between the two implementations."""
    r = sfm_ref.CvRNG()
    assert r.next() == 130063605
    for count in (6, 7, 50, 4000, 100000):
        out = np.zeros((200, 5), dtype=np.int32)
        tvh.tvh_rng(0xFFFFFFFFFFFFFFFF, count, 200, out.ctypes.data)
        r = sfm_ref.CvRNG()
        want = []
        for _ in range(200):
            idx = []
            while len(idx) < 5:
                i = r.uniform(0, count)
                while i in idx:
                    i = r.uniform(0, count)
                idx.append(i)
            want.append(idx)
        assert out.tolist() == want and all(len(set(s)) == 5 for s in want)
    for p in (0.7, 0.99, 0.999):
        for ep in (0.0, 1e-9, 0.05, 0.3, 0.5, 0.9, 0.999, 1.0):
            for mx in (1, 7, 1000):
                assert tvh.tvh_update_iters(p, ep, 5, mx) == sfm_ref.ransac_update_num_iters(p, ep, 5, mx), (p, ep, mx)
    assert sfm_ref.ransac_update_num_iters(0.7, 0.1, 5, 1000) == 1 and sfm_ref.ransac_update_num_iters(0.999, 0.5, 5, 1000) == 218


def test_decompose_and_error(tvh):
    rng = np.random.default_rng(5)
    for _ in range(50):
        R, t = _rot(rng.uniform(-1, 1, 3)), rng.uniform(-1, 1, 3)
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E = np.ascontiguousarray(tx @ R * rng.uniform(0.1, 5) + rng.normal(0, 1e-3, (3, 3)))   # an ESTIMATED E: not exactly rank two
        R1, R2, tt = np.zeros(9), np.zeros(9), np.zeros(3)
        assert tvh.tvh_decompose(E.ctypes.data, R1.ctypes.data, R2.ctypes.data, tt.ctypes.data) == 1
        w1, w2, wt = sfm_ref.decompose_essential(E)
        R1, R2 = R1.reshape(3, 3), R2.reshape(3, 3)
        same = np.allclose(R1, w1, atol=1e-9) and np.allclose(R2, w2, atol=1e-9)
        swapped = np.allclose(R1, w2, atol=1e-9) and np.allclose(R2, w1, atol=1e-9)
        assert (same or swapped) and min(np.abs(tt - wt).max(), np.abs(tt + wt).max()) < 1e-9
        assert abs(np.linalg.det(R1) - 1) < 1e-12 and abs(np.linalg.det(R2) - 1) < 1e-12
        x1, x2 = rng.uniform(-0.5, 0.5, (300, 2)), rng.uniform(-0.5, 0.5, (300, 2))
        err = np.zeros(300, dtype=np.float32)
        tvh.tvh_errors(E.ctypes.data, np.ascontiguousarray(x1).ctypes.data, np.ascontiguousarray(x2).ctypes.data, 300, err.ctypes.data)
        want = sfm_ref.sampson_errors(E, x1, x2)
        assert np.allclose(err, want, rtol=3e-7, atol=0)


def _scene(seed, n, outliers, noise=0.3):
    rng = np.random.default_rng(seed)
    camera = (1400.0, 1380.0, 960.0, 540.0)
    R, t = _rot([0.03, np.deg2rad(6.0), -0.01]), np.array([-0.6, 0.02, 0.05])
    X = np.stack([rng.uniform(-2.2, 2.2, n), rng.uniform(-1.2, 1.2, n), rng.uniform(4, 9, n)], axis=1)
    K = np.array([[camera[0], 0, camera[2]], [0, camera[1], camera[3]], [0, 0, 1.0]])
    p1, p2 = X @ K.T, (X @ R.T + t) @ K.T
    uv1 = (p1[:, :2] / p1[:, 2:3] + rng.normal(0, noise, (n, 2))).astype(np.float32)
    uv2 = (p2[:, :2] / p2[:, 2:3] + rng.normal(0, noise, (n, 2))).astype(np.float32)
    wrong = rng.uniform(size=n) < outliers
    uv2[wrong] = np.stack([rng.uniform(0, 1920, int(wrong.sum())), rng.uniform(0, 1080, int(wrong.sum()))], axis=1).astype(np.float32)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    return camera, uv1, uv2, wrong, sfm_ref.canonical_sign(tx @ R / np.linalg.norm(tx @ R))


@pytest.mark.parametrize("seed,n,outliers,prob", [(1, 800, 0.3, 0.7), (2, 800, 0.3, 0.999), (3, 3000, 0.1, 0.7), (4, 60, 0.5, 0.99),
                                                  (5, 6, 0.0, 0.7), (6, 5, 0.0, 0.7), (7, 400, 0.8, 0.9)])
def test_ransac_loop_against_the_restatement(tvh, seed, n, outliers, prob):
    """findEssentialMat(coords1, coords2, K, RANSAC, prob, 1.0, mask) as SfMUtil.cpp:39 calls it (prob 0.7) and with other confidences: the
    kernel's control flow on the host (sixteen samples per round, models replayed in the reference's order) and the sequential numpy loop
    make the same decisions -- same iterations, same inlier mask, the same E to rounding."""
    camera, uv1, uv2, wrong, Et = _scene(seed, n, outliers)
    E, mask, it = sfm_ref.find_essential_mat(uv1, uv2, camera, prob, 1.0)
    Eh, mh, ith = np.zeros(9), np.zeros(n, dtype=np.uint8), C.c_int(0)
    cam = np.array(camera)
    good = tvh.tvh_find_essential(np.ascontiguousarray(uv1).ctypes.data, np.ascontiguousarray(uv2).ctypes.data, n, cam.ctypes.data, prob, 1.0,
                                  1000, Eh.ctypes.data, mh.ctypes.data, C.byref(ith))
    assert ith.value == it and good == int(mask.sum()) and np.array_equal(mh, mask)
    if E is None:
        assert good == 0 and not Eh.any()
    else:
        assert np.abs(Eh.reshape(3, 3) - E).max() < 1e-9
        if outliers <= 0.5 and n > 50:     # the estimate is the scene's geometry: most true correspondences are inliers, few wrong ones are
            assert mask[~wrong].mean() > 0.8 and mask[wrong].mean() < 0.1 and np.abs(E - Et).max() < 0.05
