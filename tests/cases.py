"""Shared test inputs for the GMS path: seeded random pairs plus the adversarial cases SURVEY.md section 4
lists (cell borders, half-cell shifts, arg-max ties, threshold ties, empty/border cells, degenerate sizes,
index permutations, flag combinations). Everything is deterministic."""
import importlib

import numpy as np

synth = importlib.import_module("sfm-gms_amd.synth")
types = importlib.import_module("sfm-gms_amd.types")

FLAGS = [(False, False), (True, False), (False, True), (True, True)]


def _pair(xy1, xy2, query, train, size1, size2):
    kp1, kp2 = synth.make_keypoints(xy1), synth.make_keypoints(xy2)
    m = np.zeros(len(query), dtype=types.DMATCH_DTYPE)
    m["queryIdx"], m["trainIdx"] = query, train
    m["imgIdx"] = np.arange(len(query)) % 7
    m["distance"] = (np.arange(len(query)) * 0.37).astype(np.float32)
    return dict(size1=size1, size2=size2, kp1=kp1, kp2=kp2, matches=m)


def lattice(cells, size1=(1000, 1000), size2=(1000, 1000), right_grid=20, jitter=0.25, seed=3):
    """Matches placed by cell: cells = [(left_cell, right_cell, count), ...] on the 20x20 left grid and a
    right_grid x right_grid right grid; points sit near cell centres (inside the unshifted cell)."""
    rng = np.random.default_rng(seed)
    xy1, xy2 = [], []
    for lc, rc, n in cells:
        lx, ly = lc % 20, lc // 20
        rx, ry = rc % right_grid, rc // right_grid
        for _ in range(n):
            jx, jy = rng.uniform(-jitter, jitter, 2)
            xy1.append(((lx + 0.5 + jx * 0.4) * size1[0] / 20.0, (ly + 0.5 + jy * 0.4) * size1[1] / 20.0))
            xy2.append(((rx + 0.5 + jx * 0.4) * size2[0] / right_grid, (ry + 0.5 + jy * 0.4) * size2[1] / right_grid))
    n = len(xy1)
    return _pair(np.array(xy1, dtype=np.float32).reshape(-1, 2), np.array(xy2, dtype=np.float32).reshape(-1, 2),
                 np.arange(n), np.arange(n), size1, size2)


def random_pair(case_id, n=2000, size1=(1920, 1080), size2=None, **kw):
    kp1, kp2, m = synth.make_pair(case_id, size1=size1, size2=size2, n1=n, **kw)
    return dict(size1=size1, size2=size2 or size1, kp1=kp1, kp2=kp2, matches=m)


def adversarial_cases():
    """name -> case dict. All inside the parity domain."""
    out = {}
    w, h = 1000, 800
    # -- coordinates exactly on cell borders and on half-cell borders (the +0.5 shifted grids) -------------
    ks = np.arange(0, 20)
    xs = np.concatenate([ks * w / 20.0, (ks + 0.5) * w / 20.0, (ks + 0.5) * w / 20.0 - 1e-3])
    ys = np.concatenate([ks * h / 20.0, (ks + 0.5) * h / 20.0, (ks + 0.5) * h / 20.0 + 1e-3])
    gx, gy = np.meshgrid(xs, ys)
    xy = np.stack([gx.ravel(), gy.ravel()], axis=1).astype(np.float32)
    xy = xy[(xy[:, 0] < w) & (xy[:, 1] < h)]
    out["cell_borders"] = _pair(xy, xy.copy(), np.arange(len(xy)), np.arange(len(xy)), (w, h), (w, h))
    # -- points in the last half cell: rejected (x >= 20 / y >= 20) only under the shifted grid types ------
    rng = np.random.default_rng(17)
    n = 1500
    xy = np.stack([rng.uniform(0.95 * w, w - 0.01, n), rng.uniform(0.9 * h, h - 0.01, n)], axis=1).astype(np.float32)
    xy[: n // 2, 1] = rng.uniform(0, h - 0.01, n // 2)
    out["last_half_cell"] = _pair(xy, xy.copy(), np.arange(n), np.arange(n), (w, h), (w, h))
    # -- arg-max tie: two right cells with the same count; the lower index must win ----------------------------
    cells = []
    for lc in (105, 106, 107, 125, 126, 127, 145, 146, 147):
        cells += [(lc, lc + 40, 12), (lc, lc - 40, 12)]  # tie between right cells lc-40 and lc+40
    out["argmax_tie"] = lattice(cells, jitter=0.0)  # exact centres: every grid type sees the same tie
    # -- threshold tie: thresh == score exactly (kept): 9 cells x 4 matches, score 12 = 6 * sqrt(36 / 9) -----
    cells = [(210, 210, 4)]
    for lc in (189, 190, 191, 209, 211, 229, 230, 231):
        cells += [(lc, lc, 1), (lc, 5, 3)]  # 1 of 4 agrees with the motion: score = 4 + 8 = 12, T = 36
    out["thresh_tie"] = lattice(cells)
    # -- one less agreeing match: 6 * sqrt(36 / 9) = 12 > 11, the centre cell is rejected ------------------------
    cells = [(210, 210, 4)]
    for t, lc in enumerate((189, 190, 191, 209, 211, 229, 230, 231)):
        cells += [(lc, lc, 1 if t else 0), (lc, 5, 3 if t else 4)]
    out["thresh_just_below"] = lattice([c for c in cells if c[2] > 0])
    # -- corner and edge cells (numpair 4 / 6) and isolated cells (empty neighbours) ----------------------------
    cells = [(0, 0, 30), (1, 1, 30), (20, 20, 30), (21, 21, 30), (19, 19, 40), (399, 399, 40), (398, 398, 25),
             (379, 379, 25), (10, 10, 50), (200, 200, 8), (388, 388, 50)]
    out["corners_edges"] = lattice(cells)
    # -- every match in one cell; and in one cell with many distinct right cells -----------------------------------
    out["one_cell"] = lattice([(210, 133, 1800)])
    out["one_cell_scattered"] = lattice([(210, r, 3) for r in range(0, 400)])
    # -- duplicate trainIdx and permuted / repeated queryIdx (queryIdx != i) -----------------------------------------
    c = random_pair(31, n=1500, inlier_frac=0.6)
    perm = np.random.default_rng(5).permutation(len(c["matches"]))
    m = c["matches"][perm].copy()
    m["trainIdx"][::7] = m["trainIdx"][0]
    m = np.concatenate([m, m[:200]])  # the same match twice
    out["permuted_duplicates"] = dict(c, matches=m)
    # -- different image sizes, non-square, rotated and scaled ground-truth motion ---------------------------------
    out["sizes_differ"] = random_pair(32, n=2500, size1=(1280, 720), size2=(900, 1200), inlier_frac=0.7, theta_deg=90.0,
                                      scale=0.7)
    out["rot45_scale_sqrt2"] = random_pair(33, n=3000, inlier_frac=0.6, theta_deg=45.0, scale=2 ** 0.5)
    out["rot180"] = random_pair(34, n=3000, inlier_frac=0.5, theta_deg=180.0)
    out["scale_half"] = random_pair(35, n=3000, inlier_frac=0.8, scale=0.5)
    # -- nothing survives (pure outliers), so rotation/scale selection never fires: output empty -------------------
    out["all_outliers"] = random_pair(36, n=800, inlier_frac=0.0)
    # -- degenerate sizes -------------------------------------------------------------------------------------------
    c = random_pair(37, n=64, inlier_frac=1.0)
    out["m1"] = dict(c, matches=c["matches"][:1].copy())
    out["m0"] = dict(c, matches=c["matches"][:0].copy())
    # -- M > N1 and M < N1, ragged ------------------------------------------------------------------------------------
    c = random_pair(38, n=1000, inlier_frac=0.7)
    out["m_lt_n"] = dict(c, matches=c["matches"][::3].copy())
    # -- exact zeros, negative zero and the largest in-domain coordinates ----------------------------------------------
    xy = np.array([[0.0, 0.0], [-0.0, 5.0], [5.0, -0.0], [w - 1.0, h - 1.0], [w - 0.001, 3.0], [3.0, h - 0.001]] * 20,
                  dtype=np.float32)
    out["zeros_and_edges"] = _pair(xy, xy.copy(), np.arange(len(xy)), np.arange(len(xy)), (w, h), (w, h))
    return out


def domain_error_cases():
    """Inputs the reference has undefined behaviour on: must come back as GMS_ERR_DOMAIN on both sides."""
    out = {}
    c = random_pair(41, n=300, inlier_frac=0.5)
    m = c["matches"].copy()
    m["trainIdx"][5] = 300
    out["train_oob"] = dict(c, matches=m)
    m = c["matches"].copy()
    m["queryIdx"][7] = -1
    out["query_negative"] = dict(c, matches=m)
    kp = c["kp1"].copy()
    kp["x"][3] = -2.0
    out["negative_coord"] = dict(c, kp1=kp)
    kp = c["kp2"].copy()
    kp["y"][c["matches"]["trainIdx"][0]] = np.nan
    out["nan_coord"] = dict(c, kp2=kp)
    kp = c["kp2"].copy()
    kp["y"][c["matches"]["trainIdx"][1]] = 1080.0 * 1.5  # right cell beyond the right grid
    out["right_cell_oob"] = dict(c, kp2=kp)
    return out
