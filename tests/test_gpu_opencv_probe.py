"""-m gpu: SURVEY.md 8c's run-time probe. If the box this suite runs on has an OpenCV with the xfeatures2d module, its
cv2.xfeatures2d.matchGMS is the reference's own function (opencv_contrib): run it on the golden set and compare the oracle (and the HIP
path) with it. Nothing is installed or assumed: without such an OpenCV the test skips and says what it looked for."""
import numpy as np
import pytest

import golden_util

pytestmark = pytest.mark.gpu


def opencv_matchgms():
    """(callable or None, what was found)"""
    try:
        import cv2  # noqa: F401
    except Exception as e:  # noqa: BLE001
        return None, f"no cv2 module on this box ({type(e).__name__})"
    import cv2
    x = getattr(cv2, "xfeatures2d", None)
    if x is None or not hasattr(x, "matchGMS"):
        return None, f"cv2 {cv2.__version__} without xfeatures2d.matchGMS (no opencv_contrib build)"
    return x.matchGMS, f"cv2 {cv2.__version__} with xfeatures2d.matchGMS"


def test_probe_is_reported():
    fn, what = opencv_matchgms()
    print("opencv_on_box:", what)
    assert isinstance(what, str) and what


@pytest.mark.parametrize("name", golden_util.names())
def test_real_opencv_matchgms_against_oracle_and_hip(ctx, oracle, name):
    fn, what = opencv_matchgms()
    if fn is None:
        pytest.skip(what)
    import cv2
    c, _ = golden_util.load(name)
    kp1 = [cv2.KeyPoint(float(k["x"]), float(k["y"]), float(k["size"])) for k in c["kp1"]]
    kp2 = [cv2.KeyPoint(float(k["x"]), float(k["y"]), float(k["size"])) for k in c["kp2"]]
    ms = [cv2.DMatch(int(m["queryIdx"]), int(m["trainIdx"]), int(m["imgIdx"]), float(m["distance"])) for m in c["matches"]]
    for rot in (False, True):
        for scale in (False, True):
            got = fn(c["size1"], c["size2"], kp1, kp2, ms, withRotation=rot, withScale=scale, thresholdFactor=6.0)
            pairs_cv = [(d.queryIdx, d.trainIdx, d.imgIdx) for d in got]
            rc, want, _, _ = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0)
            assert rc == 0
            assert pairs_cv == [(int(m["queryIdx"]), int(m["trainIdx"]), int(m["imgIdx"])) for m in want], (name, rot, scale, what)
            out = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0)
            assert out.tobytes() == want.tobytes()
