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
    (R, t, n_good, mask uint8 255 / 0). Points are normalised with the camera matrix only; each of the four candidate poses is
    tried by triangulation (positive depth below dist_thresh in both cameras); the first with the most points wins."""
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
    return poses[w][0], poses[w][1], good[w], (masks[w] * 255).astype(np.uint8)
