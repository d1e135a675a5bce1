"""Seeded synthetic inputs for the GMS path (SURVEY.md section 8d "synthetic input recipe").

The reference feeds matchGMS with detector keypoints and BFMatcher output (FeatureMatchUtil.cpp:58-68:
M = N1 matches, queryIdx = i). No detector/matcher exists offline, so inputs of the same shape are
synthesised: keypoints uniform in the image, a ground-truth similarity motion, a fraction of true
correspondences plus uniformly random outliers.
"""
import numpy as np

from .types import DMATCH_DTYPE, KEYPOINT_DTYPE

SEED_BASE = 0x5F3759DF


def rng_for(case_id):
    return np.random.Generator(np.random.PCG64(SEED_BASE ^ int(case_id)))


def make_keypoints(xy):
    xy = np.asarray(xy, dtype=np.float32).reshape(-1, 2)
    kp = np.zeros(len(xy), dtype=KEYPOINT_DTYPE)
    kp["x"], kp["y"] = xy[:, 0], xy[:, 1]
    kp["size"], kp["angle"], kp["response"], kp["octave"], kp["class_id"] = 31.0, -1.0, 0.0, 0, -1
    return kp


def make_matches(query, train, rng=None):
    m = np.zeros(len(query), dtype=DMATCH_DTYPE)
    m["queryIdx"], m["trainIdx"] = query, train
    m["imgIdx"] = 0
    m["distance"] = (rng.uniform(0, 256, len(query)) if rng is not None else 0).astype(np.float32) \
        if rng is not None else 0
    return m


def similarity(xy, size_from, size_to, theta_deg=0.0, scale=1.0, shift=(0.0, 0.0)):
    """Rotate by theta about the source image centre, scale, move to the target centre, shift."""
    t = np.deg2rad(theta_deg)
    c, s = np.cos(t), np.sin(t)
    ctr_a = np.array([size_from[0] / 2.0, size_from[1] / 2.0])
    ctr_b = np.array([size_to[0] / 2.0, size_to[1] / 2.0])
    d = xy.astype(np.float64) - ctr_a
    r = np.stack([c * d[:, 0] - s * d[:, 1], s * d[:, 0] + c * d[:, 1]], axis=1) * scale
    return r + ctr_b + np.asarray(shift, dtype=np.float64)


def make_pair(case_id, size1=(1920, 1080), size2=None, n1=10000, n2=None, inlier_frac=0.5, theta_deg=0.0,
              scale=1.0, shift=(3.0, -2.0), noise_px=2.0):
    """One image pair: kp1, kp2, matches (M = n1, queryIdx = i) -- BASELINE configs 1, 2 and 4."""
    size2 = size2 or size1
    n2 = n2 or n1
    rng = rng_for(case_id)
    w1, h1 = size1
    w2, h2 = size2
    xy1 = np.stack([rng.uniform(0, w1 - 1, n1), rng.uniform(0, h1 - 1, n1)], axis=1).astype(np.float32)
    xy2 = np.stack([rng.uniform(0, w2 - 1, n2), rng.uniform(0, h2 - 1, n2)], axis=1).astype(np.float32)
    is_in = rng.uniform(size=n1) < inlier_frac
    train = rng.integers(0, n2, n1)
    # true correspondences overwrite slots of kp2 (distinct slots, so kp2 stays a proper keypoint set)
    slots = rng.permutation(n2)[: min(n1, n2)]
    mapped = similarity(xy1, size1, size2, theta_deg, scale, shift) + rng.normal(0, noise_px, (n1, 2))
    inb = (mapped[:, 0] >= 0) & (mapped[:, 0] < w2 - 1) & (mapped[:, 1] >= 0) & (mapped[:, 1] < h2 - 1)
    k = 0
    for i in np.nonzero(is_in & inb)[0]:
        if k >= len(slots):
            break
        xy2[slots[k]] = mapped[i]
        train[i] = slots[k]
        k += 1
    kp1, kp2 = make_keypoints(xy1), make_keypoints(xy2)
    matches = make_matches(np.arange(n1), train, rng)
    return kp1, kp2, matches


def make_sequence(case_id, n_frames, size=(1920, 1080), n_kp=10000, drift_px=6.0, noise_px=1.5, spatial_order=False):
    """A sequence of frames observing one scene under a slow drift (BASELINE config 3): frame f sees
    base point i at base[i] + f * drift + noise (kept inside the image), so keypoint i of frame a
    truly corresponds to keypoint i of frame b."""
    rng = rng_for(case_id)
    w, h = size
    margin = 0.15
    base = np.stack([rng.uniform(margin * w, (1 - margin) * w, n_kp),
                     rng.uniform(margin * h, (1 - margin) * h, n_kp)], axis=1)
    if spatial_order:  # keypoint index follows the 20 x 20 cell (a detector that emits row by row): neighbours in the list share cells
        cell = (base[:, 1] * 20 // h).astype(np.int64) * 20 + (base[:, 0] * 20 // w).astype(np.int64)
        base = base[np.argsort(cell, kind="stable")]
    direction = rng.uniform(-1, 1, 2)
    direction /= np.linalg.norm(direction)
    frames = []
    for f in range(n_frames):
        step = (f - (n_frames - 1) / 2.0) * drift_px / max(1.0, n_frames / 16.0)
        xy = base + direction * step + rng.normal(0, noise_px, (n_kp, 2))
        xy[:, 0] = np.clip(xy[:, 0], 0, w - 1.001)
        xy[:, 1] = np.clip(xy[:, 1], 0, h - 1.001)
        frames.append(make_keypoints(xy.astype(np.float32)))
    return frames


def sequence_matches(case_id, n_kp_a, n_kp_b, inlier_frac=0.5):
    """Putative matches of one pair of a make_sequence() sequence: M = n_kp_a, queryIdx = i, a fraction
    true (trainIdx = i), the rest uniformly random."""
    rng = rng_for(case_id)
    q = np.arange(n_kp_a)
    t = np.where(rng.uniform(size=n_kp_a) < inlier_frac, np.minimum(q, n_kp_b - 1), rng.integers(0, n_kp_b, n_kp_a))
    return make_matches(q, t, rng)


def sequence_descriptors(case_id, n_frames, n_kp, kind="orb", flip_bits=20, noise=6.0, outlier_frac=0.0):
    """Descriptors for a make_sequence() sequence: scene point i has one base descriptor, frame f observes it with noise, so
    the brute-force match of keypoint i of frame a in frame b is (mostly) keypoint i.
    kind "orb": uint8 [n_kp, 32] rows, `flip_bits` random bits flipped per observation (NORM_HAMMING).
    kind "sift": float32 [n_kp, 128] rows holding integers 0..255, as cv::SIFT emits them (NORM_L2).
    outlier_frac: share of a frame's keypoints that show something else (a random descriptor): their nearest neighbour in
    another frame is arbitrary, which is what the putative matches of real image pairs look like."""
    rng = rng_for(case_id ^ 0xD35C)
    out = []
    if kind == "orb":
        base = rng.integers(0, 256, (n_kp, 32), dtype=np.uint8)
        for _ in range(n_frames):
            d = base.copy()
            pos = rng.integers(0, 256, (n_kp, flip_bits))
            rows = np.repeat(np.arange(n_kp), flip_bits)
            np.bitwise_xor.at(d, (rows, (pos >> 3).ravel()), (1 << (pos & 7)).astype(np.uint8).ravel())
            other = rng.uniform(size=n_kp) < outlier_frac
            d[other] = rng.integers(0, 256, (int(other.sum()), 32), dtype=np.uint8)
            out.append(d)
    else:
        base = rng.gamma(1.2, 22.0, (n_kp, 128))
        for _ in range(n_frames):
            d = np.clip(np.rint(base + rng.normal(0, noise, (n_kp, 128))), 0, 255)
            other = rng.uniform(size=n_kp) < outlier_frac
            d[other] = np.clip(np.rint(rng.gamma(1.2, 22.0, (int(other.sum()), 128))), 0, 255)
            out.append(d.astype(np.float32))
    return out


def make_two_view_scene(case_id, size=(2016, 1512), n_points=8000, camera=(1700.0, 1690.0, 1008.0, 756.0),
                        dist=(-0.12, 0.05, 0.001, -0.0007, 0.01), yaw_deg=6.0, t=(-0.6, 0.02, 0.05), depth=(4.0, 9.0),
                        noise_px=0.3, kind="orb", descriptor_outliers=0.3):
    """Two calibrated views of one rigid scene, as structureFromMotion meets them (SfMUtil.cpp:4: two images, cameraMatrix,
    distCoeffs): 3-D points in front of camera 1 = [I|0], camera 2 = [R|t], projected through the forward distortion model
    (k1, k2, p1, p2, k3) and the camera matrix; points that leave either image are dropped. Keypoint i of frame 0 and keypoint i of
    frame 1 show the same scene point; the descriptors say so for a share 1 - descriptor_outliers of the keypoints.
    Returns dict(frames=[kp1, kp2], descriptors=[d1, d2], desc_kind, sizes, camera, dist, R, t, X)."""
    rng = rng_for(case_id ^ 0x2F1E)
    w, h = size
    fx, fy, cx, cy = camera
    half_w, half_h = 0.5 * w / fx * depth[1] * 0.55, 0.5 * h / fy * depth[1] * 0.55
    X = np.stack([rng.uniform(-half_w, half_w, n_points), rng.uniform(-half_h, half_h, n_points), rng.uniform(depth[0], depth[1], n_points)], axis=1)
    a = np.deg2rad(yaw_deg)
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ \
        np.array([[1, 0, 0], [0, np.cos(0.02), -np.sin(0.02)], [0, np.sin(0.02), np.cos(0.02)]])
    tv = np.asarray(t, dtype=np.float64)

    def project(Xc):
        x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
        k1, k2, p1, p2, k3 = dist if dist is not None else (0, 0, 0, 0, 0)
        r2 = x * x + y * y
        rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        return np.stack([xd * fx + cx, yd * fy + cy], axis=1)

    uv1, uv2 = project(X), project(X @ R.T + tv)
    uv1 += rng.normal(0, noise_px, uv1.shape)
    uv2 += rng.normal(0, noise_px, uv2.shape)
    ok = (uv1[:, 0] > 1) & (uv1[:, 0] < w - 2) & (uv1[:, 1] > 1) & (uv1[:, 1] < h - 2) & \
         (uv2[:, 0] > 1) & (uv2[:, 0] < w - 2) & (uv2[:, 1] > 1) & (uv2[:, 1] < h - 2)
    uv1, uv2, X = uv1[ok].astype(np.float32), uv2[ok].astype(np.float32), X[ok]
    n = len(uv1)
    descs = sequence_descriptors(case_id, 2, n, kind, outlier_frac=descriptor_outliers)
    return dict(frames=[make_keypoints(uv1), make_keypoints(uv2)], descriptors=descs, desc_kind=0 if kind == "orb" else 1,
                sizes=[size, size], camera=camera, dist=dist, R=R, t=tv, X=X)


def make_multi_view_scene(case_id, n_frames, size=(1920, 1080), n_kp=10000, camera=(1400.0, 1380.0, 960.0, 540.0), dist=None,
                          yaw_span_deg=8.0, x_span=1.0, depth=(4.0, 9.0), noise_px=0.3):
    """n_frames calibrated views of one rigid scene from cameras strung along a short arc (yaw and sideways translation growing with the
    frame index): exactly n_kp scene points that stay inside every image, keypoint i of every frame showing scene point i -- the input of
    structureFromMotion (SfMUtil.cpp:4) for every pair of a sequence. Returns dict(frames, sizes, camera, dist, R [n, 3, 3], t [n, 3], X)."""
    rng = rng_for(case_id ^ 0x3A7C)
    w, h = size
    fx, fy, cx, cy = camera
    Rs, ts = [], []
    for f in range(n_frames):
        s = (f / max(1, n_frames - 1)) - 0.5
        a = np.deg2rad(yaw_span_deg) * s
        Rs.append(np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]))
        ts.append(np.array([-x_span * s, 0.03 * s, 0.05 * s]))
    half_w, half_h = 0.5 * w / fx * depth[0] * 0.8, 0.5 * h / fy * depth[0] * 0.8
    k1, k2, p1, p2, k3 = dist if dist is not None else (0, 0, 0, 0, 0)

    def project(Xc):
        x, y = Xc[:, 0] / Xc[:, 2], Xc[:, 1] / Xc[:, 2]
        r2 = x * x + y * y
        rad = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        return np.stack([(x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)) * fx + cx, (y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y) * fy + cy], axis=1)

    X = np.zeros((0, 3))
    while len(X) < n_kp:
        cand = np.stack([rng.uniform(-half_w * 1.6, half_w * 1.6, 2 * n_kp), rng.uniform(-half_h * 1.4, half_h * 1.4, 2 * n_kp),
                         rng.uniform(depth[0], depth[1], 2 * n_kp)], axis=1)
        ok = np.ones(len(cand), dtype=bool)
        for R, t in zip(Rs, ts):
            uv = project(cand @ R.T + t)
            ok &= (uv[:, 0] > 2) & (uv[:, 0] < w - 3) & (uv[:, 1] > 2) & (uv[:, 1] < h - 3)
        X = np.concatenate([X, cand[ok]])
    X = X[:n_kp]
    frames = [make_keypoints((project(X @ R.T + t) + rng.normal(0, noise_px, (n_kp, 2))).astype(np.float32)) for R, t in zip(Rs, ts)]
    return dict(frames=frames, sizes=[size] * n_frames, camera=camera, dist=dist, R=np.array(Rs), t=np.array(ts), X=X)


def make_zoom_sequence(case_id, n_frames, size=(1920, 1080), n_kp=10000, zoom=2.0 ** 0.5, noise_px=1.5):
    """A sequence whose odd frames see the scene of the even frames magnified by `zoom` about the image centre (a camera that zooms
    between shots): a pair (even frame, odd frame) has true relative scale `zoom`, so with scale hypotheses the reference's winner is
    the matching right grid (sqrt 2 -> 28 x 28), not the unscaled one. Keypoint i of every frame shows scene point i."""
    rng = rng_for(case_id ^ 0x51A7)
    w, h = size
    half_w, half_h = 0.5 * w / zoom * 0.95, 0.5 * h / zoom * 0.95
    base = np.stack([rng.uniform(-half_w, half_w, n_kp), rng.uniform(-half_h, half_h, n_kp)], axis=1)
    frames = []
    for f in range(n_frames):
        s = zoom if f & 1 else 1.0
        xy = base * s + np.array([w / 2.0, h / 2.0]) + rng.normal(0, noise_px, (n_kp, 2)) + np.array([0.3 * f, -0.2 * f])
        xy[:, 0] = np.clip(xy[:, 0], 0, w - 1.001)
        xy[:, 1] = np.clip(xy[:, 1], 0, h - 1.001)
        frames.append(make_keypoints(xy.astype(np.float32)))
    return frames


def make_textured_images(case_id, n_images, size=(1920, 1080), n_boxes=9000, noise=3):
    """8-bit grey test images with corners to find: overlapping random rectangles of random grey levels plus a little noise;
    image i + 1 is image i's scene shifted by a few pixels (a sequence, not independent frames). Returns uint8 [n, H, W]."""
    rng = rng_for(case_id ^ 0x1A6E)
    w, h = size
    pad = 4 * n_images + 8
    canvas = np.full((h + pad, w + pad), 128, dtype=np.int16)
    x0 = rng.integers(0, w + pad - 4, n_boxes)
    y0 = rng.integers(0, h + pad - 4, n_boxes)
    bw = rng.integers(6, 60, n_boxes)
    bh = rng.integers(6, 60, n_boxes)
    lv = rng.integers(20, 236, n_boxes)
    for i in range(n_boxes):
        canvas[y0[i]:y0[i] + bh[i], x0[i]:x0[i] + bw[i]] = lv[i]
    out = np.empty((n_images, h, w), dtype=np.uint8)
    for i in range(n_images):
        dx, dy = 3 * i, i
        view = canvas[dy:dy + h, dx:dx + w] + rng.integers(-noise, noise + 1, (h, w), dtype=np.int16)
        out[i] = np.clip(view, 0, 255).astype(np.uint8)
    return out
