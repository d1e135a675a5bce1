"""-m gpu: the reference's own main() GMS scenario (main.cpp:19-47 -> SIFT_matchGMS, FeatureMatchUtil.cpp:52-84) from real pixels.

main() reads SourceImages/Disparity_L.jpg / Disparity_R.jpg (1920 x 1080) and calls SIFT_matchGMS three times: on the pair as it is
(main.cpp:32), with the right image turned by 180 degrees (main.cpp:36-39) and with the right image resized to 1000 x 1000
(main.cpp:44-47: size1 != size2); the wrapper detects up to 10 000 keypoints per image, matches every left descriptor to its nearest
right one (BFMatcher without cross-check: M = N1) and calls matchGMS(size1, size2, kp1, kp2, matches, out, true, true).
Here: pixels (tests/golden/image_main_scenario_1080p.npz, made by make_main_scenario_fixture.py) -> gms_detect_batch_device (10 000
keypoints; the build's own FAST/BRIEF detector, not SIFT: DESIGN.md 4.7) -> gms_bfmatch_device -> gms_filter_device(true, true, 6.0),
every stage on the GPU through the C ABI and every stage equal to its CPU statement."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "image_main_scenario_1080p.npz")
THRESHOLD, MAX_KP = 3, 10000    # the photographs are smooth: FAST threshold 3 is where both 1080p images reach SIFT::create(10000)'s 10 000


def scenario_images():
    z = np.load(GOLDEN)
    left, right = np.ascontiguousarray(z["left"]), np.ascontiguousarray(z["right"])
    return {"normal": (left, right), "rotated_180": (left, np.ascontiguousarray(right[::-1, ::-1])), "resized_1000": (left, np.ascontiguousarray(z["right_1000"]))}


@pytest.fixture(scope="module")
def detected(ctx, oracle):
    """keypoints + rows of the four distinct images, GPU and CPU, once per module"""
    batch = importlib.import_module("sfm-gms_amd.batch")
    out = {}
    for name, (a, b) in scenario_images().items():
        for side, img in (("a", a), ("b", b)):
            key = img.shape + (int(img[::7, ::11].astype(np.int64).sum()),)
            if key not in out:
                kps, rows = batch.detect_images(ctx, img[None], THRESHOLD, MAX_KP)
                out[key] = (kps[0], rows[0]) + oracle.detect(img, THRESHOLD, MAX_KP)
    return out


def _get(detected, img):
    return detected[img.shape + (int(img[::7, ::11].astype(np.int64).sum()),)]


@pytest.mark.parametrize("name", ["normal", "rotated_180", "resized_1000"])
def test_main_scenario_pair_from_pixels(ctx, pkg, oracle, detected, name):
    batch = importlib.import_module("sfm-gms_amd.batch")
    a, b = scenario_images()[name]
    (kp1, rows1, want_kp1, want_rows1), (kp2, rows2, want_kp2, want_rows2) = _get(detected, a), _get(detected, b)
    # stage 1: the keypoint source against its definition
    assert kp1.tobytes() == want_kp1.tobytes() and rows1.tobytes() == want_rows1.tobytes()
    assert kp2.tobytes() == want_kp2.tobytes() and rows2.tobytes() == want_rows2.tobytes()
    assert len(kp1) == MAX_KP and len(kp2) > 5000
    size1, size2 = (a.shape[1], a.shape[0]), (b.shape[1], b.shape[0])
    table = batch.FrameTable(ctx, [kp1, kp2], [size1, size2])
    dt = batch.DescriptorTable(ctx, table, [rows1, rows2], pkg.GMS_DESC_HAMMING256)
    pairs = np.zeros(1, dtype=pkg.PAIR_DTYPE)
    pairs[0] = (0, 1, len(kp1), 0, 0)
    # stage 2: BFMatcher::match (FeatureMatchUtil.cpp:66-68)
    matches = batch.match_pairs(ctx, dt, pairs)
    assert matches.tobytes() == oracle.bf_match(rows1, rows2, True).tobytes()
    # stage 3: matchGMS at the wrapper's flags (FeatureMatchUtil.cpp:69), and at the defaults for comparison
    for rot, scale in ((True, True), (False, False)):
        out, res, mask = batch.filter_pairs(ctx, table, pairs, matches, rot, scale, 6.0)
        rc, want, want_mask, want_res = oracle.match(size1, size2, kp1, kp2, matches, rot, scale, 6.0)
        n = int(res["n_inliers"][0])
        assert rc == 0 and int(res["status"][0]) == 0
        assert n == len(want) and out[:n].tobytes() == want.tobytes() and np.array_equal(mask, want_mask)
        assert (int(res["best_scale"][0]), int(res["best_rot"][0])) == (int(want_res[1]), int(want_res[2]))
        if rot and scale:
            assert n > 500      # real structure survives: 3187 / 2992 / 743 matches on this fixture
            if name == "rotated_180":
                assert int(res["best_rot"][0]) == 5     # rotation pattern 5 of the DLL's table = 180 degrees
    # the one-shot entry point on the same inputs (the call the reference makes: host vectors in, host vector out)
    got = pkg.matchGMS(size1, size2, kp1, kp2, matches, True, True, 6.0)
    rc, want, _, _ = oracle.match(size1, size2, kp1, kp2, matches, True, True, 6.0)
    assert got.tobytes() == want.tobytes()
