"""PODs shared with the C ABI (include/gms.h), as numpy structured dtypes."""
import numpy as np

# cv::KeyPoint, 28 bytes (stride 0x1c at DLL@0x1800485d4); GMS reads only x, y.
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
# cv::DMatch, 16 bytes (stride 0x10 at DLL@0x180046aa3); copied verbatim to the output.
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
PAIR_DTYPE = np.dtype([("frame_a", "<i4"), ("frame_b", "<i4"), ("m", "<i4"), ("reserved", "<i4"),
                       ("match_off", "<i8")])
RESULT_DTYPE = np.dtype([("n_inliers", "<i4"), ("best_scale", "<i4"), ("best_rot", "<i4"), ("status", "<i4")])

assert KEYPOINT_DTYPE.itemsize == 28 and DMATCH_DTYPE.itemsize == 16
assert PAIR_DTYPE.itemsize == 24 and RESULT_DTYPE.itemsize == 16

GMS_OK, GMS_ERR_BAD_ARG, GMS_ERR_DOMAIN, GMS_ERR_HIP, GMS_ERR_NO_DEVICE, GMS_ERR_CAPACITY = 0, -1, -2, -3, -4, -5
GMS_ERR_NOT_RESERVED, GMS_ERR_IO, GMS_ERR_NO_MODEL = -6, -7, -8
GMS_DETECT_BORDER = 16   # include/gms.h: keypoints of gms_detect_batch_device sit at least this far from every edge


class GmsError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = int(code)
        super().__init__(f"gms error {code}: {what}")

# descriptor kinds of the brute-force matcher (include/gms.h)
GMS_DESC_HAMMING256, GMS_DESC_L2_F32X128 = 0, 1

# gms_disparity_stats (include/gms.h)
DISPARITY_STATS_DTYPE = np.dtype([("count", "<i8"), ("sum_sq", "<i8"), ("max_abs", "<i4"), ("status", "<i4")])
assert DISPARITY_STATS_DTYPE.itemsize == 24
TRIANGULATION_STATS_DTYPE = np.dtype([("sum_sq_err1", "<f8"), ("sum_sq_err2", "<f8"), ("count", "<i8"), ("behind", "<i8")])
POSE_DTYPE = np.dtype([("R", "<f8", (3, 3)), ("t", "<f8", (3,)), ("n_good", "<i4"), ("which", "<i4")])
assert POSE_DTYPE.itemsize == 104

# gms_camera / gms_two_view (include/gms.h): the batched two-view stage (SfMUtil.cpp:25-82)
CAMERA_DTYPE = np.dtype([("fx", "<f8"), ("fy", "<f8"), ("cx", "<f8"), ("cy", "<f8"), ("k1", "<f8"), ("k2", "<f8"), ("p1", "<f8"),
                         ("p2", "<f8"), ("k3", "<f8")])
TWO_VIEW_DTYPE = np.dtype([("E", "<f8", (3, 3)), ("R", "<f8", (3, 3)), ("t", "<f8", (3,)), ("sum_sq_err1", "<f8"), ("sum_sq_err2", "<f8"),
                           ("n_finite", "<i8"), ("n_behind", "<i8"), ("n_points", "<i4"), ("n_ransac", "<i4"), ("ransac_iters", "<i4"),
                           ("n_pose", "<i4"), ("pose_which", "<i4"), ("n_triangulated", "<i4"), ("status", "<i4"), ("reserved", "<i4")])
assert CAMERA_DTYPE.itemsize == 72 and TWO_VIEW_DTYPE.itemsize == 232


def make_camera(camera, dist=None):
    """(fx, fy, cx, cy) and (k1, k2, p1, p2, k3) or None -> a CAMERA_DTYPE record (cameraMatrix / distCoeffs of SfMUtil.cpp:4)."""
    c = np.zeros(1, dtype=CAMERA_DTYPE)
    c["fx"], c["fy"], c["cx"], c["cy"] = camera
    if dist is not None:
        c["k1"], c["k2"], c["p1"], c["p2"], c["k3"] = dist
    return c
