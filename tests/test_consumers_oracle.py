"""CPU: the consumers' oracle (oracle/consumer_ref.c: DisparityUtil.cpp:179-201, SfMUtil.cpp:25-35) on known answers."""
import numpy as np


def _kp(synth, xy):
    return synth.make_keypoints(np.asarray(xy, dtype=np.float32))


def test_disparity_map_and_rms_known_answers(oracle, synth, pkg):
    w, h = 8, 4
    kp1 = _kp(synth, [[1.9, 0.2], [3.0, 2.7], [1.2, 0.9], [7.99, 3.99], [5.0, 1.0]])
    kp2 = _kp(synth, [[4.5, 0.0], [0.2, 3.0], [1.0, 0.0], [7.0, 0.0], [5.0, 2.0], [300.0, 0.0]])
    m = np.zeros(5, dtype=pkg.DMATCH_DTYPE)
    m["queryIdx"] = [0, 1, 2, 3, 4]
    m["trainIdx"] = [0, 1, 2, 3, 5]          # pixel (0, 1): |1 - 4| = 3 first, then match 2 overwrites it with |1 - 1| = 0
    gt = np.full((h, w), 9, dtype=np.uint8)
    rc, disp, cnt, ssq, mx, rms = oracle.disparity(kp1, kp2, m, w, h, gt, 3)
    want = np.full((h, w), 255, dtype=np.uint8)
    want[0, 1] = 0
    want[2, 3] = 3
    want[3, 7] = 0
    want[1, 5] = (300 - 5) & 255             # int -> uchar keeps the low byte: 39
    assert rc == 0 and np.array_equal(disp, want)
    a = [abs(0 - 3), abs(3 - 3), abs(0 - 3), abs(39 - 3)]
    assert cnt == 4 and ssq == sum(x * x for x in a) and mx == 36 and rms == np.sqrt(ssq / 4.0)
    # a value of exactly 255 is indistinguishable from "no match" in the reference, too
    kp2b = _kp(synth, [[260.0, 0.0]])
    m1 = m[:1].copy()
    m1["trainIdx"] = 0
    kp1b = _kp(synth, [[5.0, 1.0]])
    rc, disp, cnt, ssq, mx, rms = oracle.disparity(kp1b, kp2b, m1, w, h, gt, 3)
    assert rc == 0 and (disp == 255).all() and cnt == 0 and np.isnan(rms)
    # outside the image / bad index: undefined in the reference, an error here
    assert oracle.disparity(_kp(synth, [[8.0, 0.0]]), kp2, m[:1], w, h, gt, 3)[0] == -2
    bad = m[:1].copy()
    bad["trainIdx"] = 6
    assert oracle.disparity(kp1, kp2, bad, w, h, gt, 3)[0] == -2


def test_gather_known_answers(oracle, synth, pkg):
    kp1 = _kp(synth, [[1.5, 2.5], [3.25, 4.0]])
    kp2 = _kp(synth, [[9.0, 8.0], [7.0, 6.5], [5.0, 4.0]])
    m = np.zeros(3, dtype=pkg.DMATCH_DTYPE)
    m["queryIdx"] = [1, 0, 1]
    m["trainIdx"] = [2, 0, 1]
    rc, c1, c2 = oracle.gather(kp1, kp2, m)
    assert rc == 0 and c1.tolist() == [[3.25, 4.0], [1.5, 2.5], [3.25, 4.0]] and c2.tolist() == [[5.0, 4.0], [9.0, 8.0], [7.0, 6.5]]


def test_recover_pose_oracle_finds_the_camera_motion():
    """oracle/sfm_ref.recover_pose on exact correspondences of a known motion: the rotation, the direction of the translation, every
    point in front of both cameras."""
    import sfm_ref
    rng = np.random.default_rng(3)
    camera = (1400.0, 1380.0, 960.0, 540.0)
    ang = np.deg2rad(-9.0)
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.5, -0.05, 0.1])
    X = np.stack([rng.uniform(-2, 2, 300), rng.uniform(-1, 1, 300), rng.uniform(4, 9, 300)], axis=1)
    K = np.array([[camera[0], 0, camera[2]], [0, camera[1], camera[3]], [0, 0, 1.0]])
    p1 = X @ K.T
    p2 = (X @ R.T + t) @ K.T
    uv1, uv2 = p1[:, :2] / p1[:, 2:3], p2[:, :2] / p2[:, 2:3]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Rr, tr, good, mask = sfm_ref.recover_pose(tx @ R, uv1, uv2, camera)
    assert good == 300 and (mask == 255).all()
    assert np.allclose(Rr, R, atol=1e-9) and np.allclose(tr, t / np.linalg.norm(t), atol=1e-9)
