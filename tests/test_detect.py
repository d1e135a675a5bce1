"""CPU: the keypoint source's definition (oracle/detect_ref.c) against an independent numpy statement of the same definition, its
pattern generator pinned by digest, and the selection rules on constructed images. The reference's detectors are OpenCV's (a binary
dependency: FeatureMatchUtil.cpp:9-12, DisparityUtil.cpp:108,123-138); this detector is the build's own, so the oracle here IS the
definition and these tests are its double-entry bookkeeping."""
import hashlib
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "image_stereo_pair_450x375.npz")
COS = np.array([4096, 4017, 3784, 3406, 2896, 2276, 1567, 799, 0, -799, -1567, -2276, -2896, -3406, -3784, -4017,
                -4096, -4017, -3784, -3406, -2896, -2276, -1567, -799, 0, 799, 1567, 2276, 2896, 3406, 3784, 4017], dtype=np.int64)
SIN = np.roll(COS, 8)
CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]


def np_score(img):
    h, w = img.shape
    I = img.astype(np.int32)
    B = 16
    core = I[B:h - B, B:w - B]
    d = np.stack([I[B + dy:h - B + dy, B + dx:w - B + dx] - core for dx, dy in CIRCLE])      # [16, h', w']
    best = np.zeros_like(core)
    for s in range(16):
        arc = d[[(s + k) % 16 for k in range(9)]]
        best = np.maximum(best, np.maximum(arc.min(axis=0), -arc.max(axis=0)))
    out = np.zeros((h, w), dtype=np.uint8)
    out[B:h - B, B:w - B] = best
    return out


def np_detect(img, threshold, max_kp, pattern):
    h, w = img.shape
    sc = np_score(img).astype(np.int32)
    nb = np.zeros_like(sc)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                nb[1:h - 1, 1:w - 1] = np.maximum(nb[1:h - 1, 1:w - 1], sc[1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx])
    cand = (sc > threshold) & (sc > nb)
    ys, xs = np.nonzero(cand)                       # raster order
    s = sc[ys, xs]
    if len(s) > max_kp:
        order = np.lexsort((np.arange(len(s)), -s))[:max_kp]     # highest scores, earlier raster position first among equals
        keep = np.sort(order)
        ys, xs, s = ys[keep], xs[keep], s[keep]
    I = img.astype(np.int64)
    box = np.zeros((h, w), dtype=np.int64)
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            box[2:h - 2, 2:w - 2] += I[2 + dy:h - 2 + dy, 2 + dx:w - 2 + dx]
    dd = np.arange(-15, 16)
    DX, DY = np.meshgrid(dd, dd)
    disc = DX * DX + DY * DY <= 225
    bins, rows = [], []
    for x, y in zip(xs, ys):
        patch = I[y - 15:y + 16, x - 15:x + 16]
        m10, m01 = int((patch * DX)[disc].sum()), int((patch * DY)[disc].sum())
        b = int(np.argmax(m10 * COS + m01 * SIN))    # first maximum
        c, sn = int(COS[b]), int(SIN[b])
        p = pattern.astype(np.int64)
        rax, ray = (p[:, 0] * c - p[:, 1] * sn + 2048) >> 12, (p[:, 0] * sn + p[:, 1] * c + 2048) >> 12
        rbx, rby = (p[:, 2] * c - p[:, 3] * sn + 2048) >> 12, (p[:, 2] * sn + p[:, 3] * c + 2048) >> 12
        bits = box[y + ray, x + rax] < box[y + rby, x + rbx]
        rows.append(np.packbits(bits, bitorder="little"))
        bins.append(b)
    return xs, ys, s, np.array(bins), (np.array(rows) if rows else np.zeros((0, 32), np.uint8))


def test_pattern_generator_is_pinned(oracle):
    pat = oracle.detect_pattern()
    assert pat.shape == (256, 4) and pat.dtype == np.int8
    assert ((pat[:, 0].astype(int) ** 2 + pat[:, 1].astype(int) ** 2) <= 144).all()
    assert ((pat[:, 2].astype(int) ** 2 + pat[:, 3].astype(int) ** 2) <= 144).all()
    assert ((pat[:, 0] != pat[:, 2]) | (pat[:, 1] != pat[:, 3])).all()
    assert hashlib.sha256(pat.tobytes()).hexdigest()[:16] == PATTERN_SHA16
    assert pat[:2].tolist() == [[-5, 10, 6, 8], [-1, 7, -3, 11]]


@pytest.mark.parametrize("which,threshold,max_kp", [("left", 20, 10000), ("right", 20, 10000), ("left", 8, 700), ("right", 40, 10000)])
def test_oracle_agrees_with_the_numpy_statement_on_the_real_pair(oracle, which, threshold, max_kp):
    img = np.load(GOLDEN)[which]
    kp, rows = oracle.detect(img, threshold, max_kp)
    xs, ys, s, bins, want_rows = np_detect(img, threshold, max_kp, oracle.detect_pattern())
    assert len(kp) == len(xs) and len(kp) > 300
    assert (kp["x"] == xs).all() and (kp["y"] == ys).all() and (kp["response"] == s).all()
    assert (kp["angle"] == 11.25 * bins).all() and (kp["size"] == 31).all() and (kp["octave"] == 0).all() and (kp["class_id"] == -1).all()
    assert rows.tobytes() == want_rows.tobytes()
    sc, box = oracle.detect_maps(img)
    assert (sc == np_score(img)).all()


def test_constructed_images(oracle):
    flat = np.full((80, 90), 77, dtype=np.uint8)
    assert len(oracle.detect(flat, 0, 100)[0]) == 0
    dot = flat.copy()
    dot[40, 45] = 255                        # an isolated bright pixel: all sixteen circle pixels 178 darker
    kp, rows = oracle.detect(dot, 20, 100)
    assert len(kp) == 1 and (kp["x"][0], kp["y"][0], kp["response"][0]) == (45, 40, 178)
    edge = flat.copy()
    edge[:, 45:] = 200                       # a straight edge: at most eight contiguous circle pixels differ -> no corner
    assert len(oracle.detect(edge, 10, 100)[0]) == 0
    near_border = flat.copy()
    near_border[15, 30] = 255                # one pixel outside the keypoint region
    near_border[16, 60] = 255                # on its first row
    kp, _ = oracle.detect(near_border, 20, 100)
    assert [(int(k["x"]), int(k["y"])) for k in kp] == [(60, 16)]
    assert len(oracle.detect(np.zeros((32, 200), np.uint8), 5, 10)[0]) == 0     # no room for a keypoint


def test_selection_keeps_the_strongest_and_breaks_ties_in_raster_order(oracle):
    img = np.full((120, 160), 50, dtype=np.uint8)
    spots = [(20, 20, 90), (30, 20, 120), (40, 25, 90), (50, 30, 90), (25, 60, 200), (70, 61, 90), (100, 90, 120)]   # (x, y, value)
    for x, y, v in spots:
        img[y, x] = v
    kp, _ = oracle.detect(img, 10, 100)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kp] == [(x, y, v - 50) for x, y, v in sorted(spots, key=lambda t: (t[1], t[0]))]
    kp, _ = oracle.detect(img, 10, 5)        # three strongest (150, 70, 70) + the first two of the four 40s, in raster order
    assert [(int(k["x"]), int(k["y"])) for k in kp] == [(20, 20), (30, 20), (40, 25), (25, 60), (100, 90)]
    kp, _ = oracle.detect(img, 10, 1)
    assert [(int(k["x"]), int(k["y"])) for k in kp] == [(25, 60)]
    kp, _ = oracle.detect(img, 70, 100)      # threshold is strict
    assert [(int(k["x"]), int(k["y"])) for k in kp] == [(25, 60)]


def test_describe_at_given_keypoints(oracle, pkg):
    img = np.load(GOLDEN)["left"]
    kp, rows = oracle.detect(img, 20, 300)
    again = kp.copy()
    again["angle"] = -1
    rc, kp2, rows2 = oracle.describe(img, again)
    assert rc == len(kp) and kp2.tobytes() == kp.tobytes() and rows2.tobytes() == rows.tobytes()
    bad = kp[:3].copy()
    bad["x"][1] = 15.0
    assert oracle.describe(img, bad)[0] == -1
    bad["x"][1] = 100.5
    assert oracle.describe(img, bad)[0] == -1


PATTERN_SHA16 = "bb92bd4e0f99466f"
