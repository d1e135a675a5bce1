"""oracle/sfm_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT. numpy (fp64) restatement of the tail of the reference's two-view pipeline,
SfM-GMS/SfM-GMS/SfMUtil.cpp:76-82 and 128-143: cv::undistortPoints -> cv::triangulatePoints -> division by the fourth coordinate.
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
