"""oracle/sfm_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT. numpy (fp64) restatement of the tail of the reference's two-view pipeline,
SfM-GMS/SfM-GMS/SfMUtil.cpp:45 (cv::recoverPose), 76-82 and 128-143: cv::undistortPoints -> cv::triangulatePoints -> division by the
fourth coordinate.
undistortPoints and triangulatePoints live in opencv_world452 (calib3d), which the reference vendors as a Windows import library
only: parity unpinned; restated from the published algorithms (five fixed-point iterations of the distortion model; per point the
right singular vector of the smallest singular value of the 4 x 4 DLT matrix). Floating point: compared at a stated tolerance."""
import numpy as np


def undistort_points(uv, camera, dist=None):
    fx, fy, cx, cy = camera
    x0 = (np.asarray(uv, dtype=np.float64)[:, 0] - cx) / fx
    y0 = (np.asarray(uv, dtype=np.float64)[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    if dist is not None and np.any(np.asarray(dist) != 0):
        k1, k2, p1, p2, k3 = dist
        for _ in range(5):
            r2 = x * x + y * y
            icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
            dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
            dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
            x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
    return np.stack([x, y], axis=1)


def triangulate(P1, P2, xy1, xy2):
    """cv::triangulatePoints + SfMUtil.cpp:131-141: [n, 3] points."""
    P1, P2 = np.asarray(P1, dtype=np.float64).reshape(3, 4), np.asarray(P2, dtype=np.float64).reshape(3, 4)
    out = np.zeros((len(xy1), 3))
    for i in range(len(xy1)):
        A = np.stack([xy1[i, 0] * P1[2] - P1[0], xy1[i, 1] * P1[2] - P1[1], xy2[i, 0] * P2[2] - P2[0], xy2[i, 1] * P2[2] - P2[1]])
        X = np.linalg.svd(A)[2][3]
        out[i] = X[:3] / X[3]
    return out


def reprojection_sums(P1, P2, xy1, xy2, pts):
    P1, P2 = np.asarray(P1, dtype=np.float64).reshape(3, 4), np.asarray(P2, dtype=np.float64).reshape(3, 4)
    h = np.concatenate([pts, np.ones((len(pts), 1))], axis=1)
    a, b = h @ P1.T, h @ P2.T
    e1 = ((a[:, :2] / a[:, 2:3] - xy1) ** 2).sum(axis=1)
    e2 = ((b[:, :2] / b[:, 2:3] - xy2) ** 2).sum(axis=1)
    return float(e1.sum()), float(e2.sum()), int(((a[:, 2] <= 0) | (b[:, 2] <= 0)).sum())


def decompose_essential(E):
    """cv::decomposeEssentialMat: (R1, R2, t) with det U = det Vt = +1, W = [0 1 0; -1 0 0; 0 0 1]."""
    U, _, Vt = np.linalg.svd(np.asarray(E, dtype=np.float64).reshape(3, 3))
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    return U @ W @ Vt, U @ W.T @ Vt, U[:, 2].copy()


def recover_pose(E, uv1, uv2, camera, in_mask=None, dist_thresh=50.0):
    """cv::recoverPose(E, points1, points2, cameraMatrix, R, t, mask) of OpenCV 4.5.2 (SfMUtil.cpp:45): returns
    (R, t, n_good, mask uint8). Points are normalised with the camera matrix only; each of the four candidate poses is
    tried by triangulation (positive depth below dist_thresh in both cameras); the first with the most points wins. The mask is what
    cv::recoverPose leaves in its in/out argument: bitwise_and(mask, hypothesis mask) -- the caller's bytes where the point passes
    (255 without an input mask), 0 elsewhere."""
    fx, fy, cx, cy = camera
    uv1, uv2 = np.asarray(uv1, dtype=np.float64), np.asarray(uv2, dtype=np.float64)
    x1 = np.stack([(uv1[:, 0] - cx) / fx, (uv1[:, 1] - cy) / fy], axis=1)
    x2 = np.stack([(uv2[:, 0] - cx) / fx, (uv2[:, 1] - cy) / fy], axis=1)
    R1, R2, t = decompose_essential(E)
    P0 = np.hstack([np.eye(3), np.zeros((3, 1))])
    masks, poses = [], [(R1, t), (R2, t), (R1, -t), (R2, -t)]
    for R, tt in poses:
        P = np.hstack([R, tt.reshape(3, 1)])
        m = np.zeros(len(x1), dtype=bool)
        for i in range(len(x1)):
            A = np.stack([x1[i, 0] * P0[2] - P0[0], x1[i, 1] * P0[2] - P0[1], x2[i, 0] * P[2] - P[0], x2[i, 1] * P[2] - P[1]])
            Q = np.linalg.svd(A)[2][3]
            ok = Q[2] * Q[3] > 0
            q = Q[:3] / Q[3]
            ok = ok and q[2] < dist_thresh
            z2 = P[2, :3] @ q + P[2, 3]
            m[i] = ok and 0 < z2 < dist_thresh
        if in_mask is not None:
            m &= np.asarray(in_mask) != 0
        masks.append(m)
    good = [int(m.sum()) for m in masks]
    if good[0] >= good[1] and good[0] >= good[2] and good[0] >= good[3]:
        w = 0
    elif good[1] >= good[0] and good[1] >= good[2] and good[1] >= good[3]:
        w = 1
    elif good[2] >= good[0] and good[2] >= good[1] and good[2] >= good[3]:
        w = 2
    else:
        w = 3
    out = np.where(masks[w], 255 if in_mask is None else np.asarray(in_mask, dtype=np.uint8), 0).astype(np.uint8)
    return poses[w][0], poses[w][1], good[w], out


def two_view(uv1, uv2, camera, dist=None, prob=0.999, threshold=1.0, max_iters=1000):
    """SfMUtil.cpp:39-82 on gathered coordinates: findEssentialMat -> recoverPose (mask in / out) -> inliers compacted in order ->
    undistortPoints -> triangulatePoints with [I|0], [R|t] -> / w. Returns a dict; E None when no model was found."""
    E, mask, iters = find_essential_mat(uv1, uv2, camera, prob, threshold, max_iters)
    out = dict(E=E, ransac_mask=mask.copy(), n_ransac=int(mask.sum()), iters=iters)
    if E is None:
        return out
    R, t, good, mask2 = recover_pose(E, uv1, uv2, camera, mask)
    keep = mask2 != 0
    xy1 = undistort_points(np.asarray(uv1)[keep], camera, dist)
    xy2 = undistort_points(np.asarray(uv2)[keep], camera, dist)
    P1, P2 = np.hstack([np.eye(3), np.zeros((3, 1))]), np.hstack([R, t.reshape(3, 1)])
    pts = triangulate(P1, P2, xy1, xy2) if keep.any() else np.zeros((0, 3))
    e1, e2, behind = reprojection_sums(P1, P2, xy1, xy2, pts) if keep.any() else (0.0, 0.0, 0)
    out.update(R=R, t=t, n_pose=good, mask=mask2, points=pts, sum_sq_err1=e1, sum_sq_err2=e2, behind=behind)
    return out


# ---- cv::findEssentialMat(points1, points2, cameraMatrix, RANSAC, prob, threshold, mask) -- SfMUtil.cpp:39 ---------------------------
# OpenCV 4.5.2 (calib3d: five-point.cpp, ptsetreg.cpp, rng of core), vendored by the reference as an import library only: PARITY
# UNPINNED. Restated from the published algorithm and the published structure of that code:
#   * points to fp64, normalised with the camera matrix ((u - cx) / fx, (v - cy) / fy); threshold /= (fx + fy) / 2;
#   * RANSACPointSetRegistrator::run with modelPoints = 5, maxIters = 1000: cv::RNG(0xFFFFFFFFFFFFFFFF) (multiply-with-carry,
#     uniform(0, n) = next() % n); a sample = five DIFFERENT indices, each drawn until it differs from the ones before;
#   * the minimal solver: Nister's five-point method -- null space of the 5 x 9 epipolar system, the ten cubic constraints
#     det E = 0, 2 E E^T E - tr(E E^T) E = 0 on E = x X + y Y + z Z + W, elimination to a 3 x 3 polynomial matrix in z, the real roots
#     of its determinant (degree 10; |imag| < 1e-10), (x, y) from the null vector of B(z); every E scaled to unit Frobenius norm;
#   * the error of a correspondence: (x2^T E x1)^2 / ((E x1)_0^2 + (E x1)_1^2 + (E^T x2)_0^2 + (E^T x2)_1^2), rounded to fp32 and
#     compared with (float)(threshold^2) by '<=';
#   * a model replaces the best one when its inlier count is strictly larger (and at least 5); the iteration bound then becomes
#     RANSACUpdateNumIters(prob, outlier ratio, 5, bound).
# Two things an SVD leaves open are fixed here by definition (and by the GPU implementation alike): every E carries the sign that
# makes its largest-magnitude entry positive, and the models of one sample are taken in ascending order of E[0][0] (then E[0][1], ...)
# -- the order decides ties between models of equal inlier count, and the roots z themselves depend on the null-space basis.
class CvRNG:
    """cv::RNG: state = (uint32)state * 4164903690 + (state >> 32); next() = (uint32)state."""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else a + self.next() % (b - a)


def ransac_update_num_iters(p, ep, model_points, max_iters):
    """cv::RANSACUpdateNumIters."""
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    return max_iters if (denom >= 0 or -num >= max_iters * (-denom)) else int(np.rint(num / denom))


# monomials of degree <= 3 in (x, y, z), in the order the elimination needs: the first ten columns are eliminated, rows 4..9 of the
# result then read  x^2 z, x^2, y^2 z, y^2, x y z, x y  = -(a polynomial in the last ten)
_MONO = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
         (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]


def _pmul(a, b):
    """product of two polynomials given as {(i, j, k): coefficient}"""
    out = {}
    for (i, j, k), u in a.items():
        for (l, m, n), v in b.items():
            key = (i + l, j + m, k + n)
            out[key] = out.get(key, 0.0) + u * v
    return out


def _padd(a, b, sb=1.0):
    out = dict(a)
    for key, v in b.items():
        out[key] = out.get(key, 0.0) + sb * v
    return out


def canonical_sign(E):
    k = int(np.argmax(np.abs(E)))
    return -E if E.reshape(-1)[k] < 0 else E


def _constraints(E):
    G = E @ E.T
    return np.concatenate([[np.linalg.det(E)], (2.0 * G @ E - np.trace(G) * E).reshape(-1)])


def _polish(basis, xyz):
    """Gauss-Newton on the ten constraints themselves (det E, 2 E E^T E - tr(E E^T) E) over (x, y, z), E = x X + y Y + z Z + W: the
    eliminated system and its degree-10 root carry the conditioning of the elimination, the constraints do not. A step (halved up to
    seven times if need be) is taken only while it lowers the residual; at most ten."""
    xyz = np.array(xyz, dtype=np.float64)
    E = np.tensordot(np.append(xyz, 1.0), basis, axes=1)
    r = _constraints(E)
    for _ in range(10):
        if not (r @ r > 0):
            break
        G, tr = E @ E.T, np.trace(E @ E.T)
        J = np.zeros((10, 3))
        for v in range(3):
            D = basis[v]
            # d det[D] = tr(adj(E) D)
            adj = np.array([[E[1, 1] * E[2, 2] - E[1, 2] * E[2, 1], E[0, 2] * E[2, 1] - E[0, 1] * E[2, 2], E[0, 1] * E[1, 2] - E[0, 2] * E[1, 1]],
                            [E[1, 2] * E[2, 0] - E[1, 0] * E[2, 2], E[0, 0] * E[2, 2] - E[0, 2] * E[2, 0], E[0, 2] * E[1, 0] - E[0, 0] * E[1, 2]],
                            [E[1, 0] * E[2, 1] - E[1, 1] * E[2, 0], E[0, 1] * E[2, 0] - E[0, 0] * E[2, 1], E[0, 0] * E[1, 1] - E[0, 1] * E[1, 0]]])
            J[0, v] = np.trace(adj @ D)
            J[1:, v] = (2.0 * (D @ E.T @ E + E @ D.T @ E + G @ D) - 2.0 * np.trace(E @ D.T) * E - tr * D).reshape(-1)
        N, g = J.T @ J, J.T @ r
        if not np.isfinite(N).all() or abs(np.linalg.det(N)) == 0:
            break
        step = -np.linalg.solve(N, g)
        if not np.isfinite(step).all():
            break
        for half in range(8):          # the full step, else halved until the residual drops
            trial = xyz + step * 0.5 ** half
            En = np.tensordot(np.append(trial, 1.0), basis, axes=1)
            rn = _constraints(En)
            if rn @ rn < r @ r:
                break
        else:
            break
        xyz, E, r = trial, En, rn
    return E


def five_point(x1, x2):
    """The essential matrices (unit Frobenius norm, largest entry positive, x2^T E x1 = 0) through five correspondences, ascending in
    E[0][0], E[0][1], ... x1, x2: [5, 2]."""
    x1, x2 = np.asarray(x1, dtype=np.float64), np.asarray(x2, dtype=np.float64)
    Q = np.stack([x2[:, 0] * x1[:, 0], x2[:, 0] * x1[:, 1], x2[:, 0], x2[:, 1] * x1[:, 0], x2[:, 1] * x1[:, 1], x2[:, 1],
                  x1[:, 0], x1[:, 1], np.ones(5)], axis=1)             # row . vec(E) (row-major) = x2^T E x1
    basis = np.linalg.svd(Q)[2][5:9].reshape(4, 3, 3)                   # X, Y, Z, W
    lin = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
    E = [[{lin[b]: basis[b, r, c] for b in range(4)} for c in range(3)] for r in range(3)]
    EEt = [[None] * 3 for _ in range(3)]
    for r in range(3):
        for c in range(3):
            acc = {}
            for k in range(3):
                acc = _padd(acc, _pmul(E[r][k], E[c][k]))
            EEt[r][c] = acc
    trace = _padd(_padd(EEt[0][0], EEt[1][1]), EEt[2][2])
    eqs = []
    det = _padd(_padd(_pmul(E[0][0], _padd(_pmul(E[1][1], E[2][2]), _pmul(E[1][2], E[2][1]), -1.0)),
                      _pmul(E[0][1], _padd(_pmul(E[1][0], E[2][2]), _pmul(E[1][2], E[2][0]), -1.0)), -1.0),
                _pmul(E[0][2], _padd(_pmul(E[1][0], E[2][1]), _pmul(E[1][1], E[2][0]), -1.0)))
    eqs.append(det)
    for r in range(3):
        for c in range(3):
            acc = {}
            for k in range(3):
                acc = _padd(acc, _pmul(EEt[r][k], E[k][c]), 2.0)
            eqs.append(_padd(acc, _pmul(trace, E[r][c]), -1.0))
    A = np.array([[eq.get(mono, 0.0) for mono in _MONO] for eq in eqs])
    try:
        G = np.linalg.solve(A[:, :10], A[:, 10:])
    except np.linalg.LinAlgError:
        return []
    B = np.zeros((3, 13))               # per row: x's cubic in z (z^3..1), y's cubic, the quartic (z^4..1)
    for i in range(3):
        a, b = G[4 + 2 * i], G[5 + 2 * i]    # (x^2 z | y^2 z | x y z) row minus z times the (x^2 | y^2 | x y) row
        for o, off in ((0, 0), (3, 4)):
            B[i, off:off + 4] = [-b[o], a[o] - b[o + 1], a[o + 1] - b[o + 2], a[o + 2]]
        B[i, 8:13] = [-b[6], a[6] - b[7], a[7] - b[8], a[8] - b[9], a[9]]
    P = lambda i, j: np.poly1d(B[i, 4 * j:4 * j + (5 if j == 2 else 4)])
    detB = (P(0, 0) * (P(1, 1) * P(2, 2) - P(1, 2) * P(2, 1)) - P(0, 1) * (P(1, 0) * P(2, 2) - P(1, 2) * P(2, 0))
            + P(0, 2) * (P(1, 0) * P(2, 1) - P(1, 1) * P(2, 0)))
    coeffs = np.zeros(11)
    coeffs[11 - len(detB.coeffs):] = detB.coeffs
    if not np.isfinite(coeffs).all() or not coeffs.any():
        return []
    roots = np.roots(coeffs)
    out = []
    for z in sorted(r.real for r in roots if abs(r.imag) < 1e-10):
        Bz = np.array([[P(i, j)(z) for j in range(3)] for i in range(3)])
        v = np.linalg.svd(Bz)[2][2]
        if abs(v[2]) < 1e-10:          # (v is a unit vector)
            continue
        Em = _polish(basis, (v[0] / v[2], v[1] / v[2], z))
        out.append(canonical_sign(Em / np.linalg.norm(Em)))
    return sorted(out, key=lambda m: tuple(m.reshape(-1)))


def sampson_errors(E, x1, x2):
    """EMEstimatorCallback::computeError: fp32 values."""
    h1 = np.concatenate([x1, np.ones((len(x1), 1))], axis=1)
    h2 = np.concatenate([x2, np.ones((len(x2), 1))], axis=1)
    Ex1, Etx2 = h1 @ E.T, h2 @ E
    num = (h2 * Ex1).sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (num * num / (Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2)).astype(np.float32)


def find_essential_mat(uv1, uv2, camera, prob=0.999, threshold=1.0, max_iters=1000, trace=None):
    """cv::findEssentialMat(points1, points2, cameraMatrix, RANSAC, prob, threshold, mask) (SfMUtil.cpp:39 passes 0.7, 1.0):
    returns (E [3, 3] or None, mask uint8 [n] of 1 / 0, iterations run). `trace`, when a list, receives (iteration, sample, models)."""
    fx, fy, cx, cy = camera
    uv1, uv2 = np.asarray(uv1, dtype=np.float64).reshape(-1, 2), np.asarray(uv2, dtype=np.float64).reshape(-1, 2)
    x1 = np.stack([(uv1[:, 0] - cx) / fx, (uv1[:, 1] - cy) / fy], axis=1)
    x2 = np.stack([(uv2[:, 0] - cx) / fx, (uv2[:, 1] - cy) / fy], axis=1)
    n = len(x1)
    thr = threshold / ((fx + fy) / 2)
    t = np.float32(thr * thr)
    if n < 5:
        return None, np.zeros(n, dtype=np.uint8), 0
    if n == 5:            # RANSACPointSetRegistrator::run: exactly a minimal sample -> its (first) model, every point an inlier
        models = five_point(x1, x2)
        return (canonical_sign(models[0]) if models else None), np.ones(n, dtype=np.uint8) * (1 if models else 0), 0
    rng = CvRNG()
    niters, best_count, best_E, best_mask = max(max_iters, 1), 0, None, np.zeros(n, dtype=np.uint8)
    it = 0
    while it < niters:
        idx = []
        while len(idx) < 5:
            i = rng.uniform(0, n)
            while i in idx:
                i = rng.uniform(0, n)
            idx.append(i)
        models = five_point(x1[idx], x2[idx])
        if trace is not None:
            trace.append((it, list(idx), models))
        for Em in models:
            mask = sampson_errors(Em, x1, x2) <= t
            good = int(mask.sum())
            if good > max(best_count, 4):
                best_count, best_E, best_mask = good, Em, mask.astype(np.uint8)
                niters = ransac_update_num_iters(prob, (n - good) / n, 5, niters)
        it += 1
    return (canonical_sign(best_E) if best_E is not None else None), best_mask, it
