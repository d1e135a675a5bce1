"""Python mirror of the reference's operator: same name, argument meaning and output as

    cv::xfeatures2d::matchGMS(size1, size2, keypoints1, keypoints2, matches1to2, matchesGMS,
                              withRotation=false, withScale=false, thresholdFactor=6.0)

(reference call sites: SfM-GMS/SfM-GMS/FeatureMatchUtil.cpp:69, DisparityUtil.cpp:149,299).
Inputs/outputs are numpy structured arrays laid out exactly like cv::KeyPoint / cv::DMatch.
All work happens in csrc/libgms_hip.so on the GPU; nothing here computes the filter.
"""
import ctypes as C

import numpy as np

from .capi import load_library
from .types import (DMATCH_DTYPE, KEYPOINT_DTYPE, PAIR_DTYPE, RESULT_DTYPE, GMS_OK, GmsError)


def _as(arr, dtype, name):
    a = np.ascontiguousarray(arr)
    if a.dtype != dtype:
        raise TypeError(f"{name} must have dtype {dtype}, got {a.dtype}")
    return a


def _check(rc, lib, what):
    if rc != GMS_OK:
        msg = lib.gms_error_string(rc).decode()
        if rc == -3:
            msg += f" (hipError {lib.gms_last_hip_error()})"
        raise GmsError(rc, f"{what}: {msg}")


class GmsContext:
    """gms_ctx: one per (process, device). Thread-safe for the one-shot call; stream-ordered batch calls."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        _check(self._lib.gms_ctx_create(int(device), C.byref(h)), self._lib, "gms_ctx_create")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gms_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def max_matches(self):
        return int(self._lib.gms_max_matches())

    def set_stream(self, hip_stream_handle):
        """Launch on a caller-owned HIP stream (e.g. torch.cuda.Stream().cuda_stream); 0/None = own stream."""
        _check(self._lib.gms_ctx_set_stream(self._h, C.c_void_p(hip_stream_handle or 0)), self._lib, "set_stream")

    def synchronize(self):
        _check(self._lib.gms_ctx_synchronize(self._h), self._lib, "synchronize")

    def set_option(self, option, value):
        """gms_ctx_set_option: option 1 = deal the matches to the lanes, 2 = probe scale hypotheses; value -1 (library's choice), 0, 1."""
        _check(self._lib.gms_ctx_set_option(self._h, int(option), int(value)), self._lib, "gms_ctx_set_option")

    def query(self, what):
        """gms_ctx_query: 1 = last launch dealt, 2 = its scale-probe mask, 3 = its matches per thread, 4 = launches, 5 = CUs, 6 / 7 = touch-ahead grid type / pairs ahead, 8 = last launch's first-round stagger (10 ns ticks)."""
        v = C.c_int64(0)
        _check(self._lib.gms_ctx_query(self._h, int(what), C.byref(v)), self._lib, "gms_ctx_query")
        return int(v.value)

    # -- one-shot, host arrays ---------------------------------------------------------------------
    def match(self, size1, size2, keypoints1, keypoints2, matches1to2, withRotation=False, withScale=False,
              thresholdFactor=6.0, return_result=False):
        kp1 = _as(keypoints1, KEYPOINT_DTYPE, "keypoints1")
        kp2 = _as(keypoints2, KEYPOINT_DTYPE, "keypoints2")
        mt = _as(matches1to2, DMATCH_DTYPE, "matches1to2")
        out = np.empty(max(len(mt), 1), dtype=DMATCH_DTYPE)
        n_out = C.c_int(0)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        rc = self._lib.gms_match_ctx(self._h, kp1.ctypes.data, len(kp1), int(size1[0]), int(size1[1]),
                                     kp2.ctypes.data, len(kp2), int(size2[0]), int(size2[1]),
                                     mt.ctypes.data, len(mt), int(bool(withRotation)), int(bool(withScale)),
                                     float(thresholdFactor), out.ctypes.data, C.byref(n_out), res.ctypes.data)
        _check(rc, self._lib, "gms_match_ctx")
        kept = out[: n_out.value].copy()
        return (kept, res[0]) if return_result else kept

    # -- host-pointer batch path: many pairs per call, pinned staging and two streams inside the library ------------
    def filter_host_batch(self, keypoints_per_frame, sizes, pairs, matches, withRotation=False, withScale=False,
                          thresholdFactor=6.0, out=None, results=None):
        """gms_filter_host_batch: `pairs` (PAIR_DTYPE) index `matches` (DMATCH_DTYPE) by match_off. Returns
        (out, results): pair i's survivors at out[match_off : match_off + results[i].n_inliers].
        keypoints_per_frame: a list of per-frame KEYPOINT_DTYPE arrays, or (all keypoints back to back, frame offsets [n_frames + 1])
        as the C ABI takes them (no concatenation per call). out / results: arrays of a previous call to write into -- a fresh
        output array costs first-touch page faults inside the call (7 ms for 327 MB), a C++ caller's std::vector has been touched
        by its constructor."""
        if isinstance(keypoints_per_frame, tuple):
            kp, frame_off = keypoints_per_frame
            kp = _as(kp, KEYPOINT_DTYPE, "keypoints") if len(kp) else np.zeros(1, dtype=KEYPOINT_DTYPE)
            frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
            n_frames = len(frame_off) - 1
        else:
            counts = np.array([len(k) for k in keypoints_per_frame], dtype=np.int64)
            frame_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            kp = (np.concatenate([_as(k, KEYPOINT_DTYPE, "keypoints") for k in keypoints_per_frame])
                  if frame_off[-1] else np.zeros(1, dtype=KEYPOINT_DTYPE))
            n_frames = len(counts)
        wh = np.ascontiguousarray(np.asarray(sizes, dtype=np.int32).reshape(-1, 2))
        if wh.shape[0] != n_frames:
            raise ValueError("one (width, height) per frame")
        pairs = _as(pairs, PAIR_DTYPE, "pairs")
        mt = _as(matches, DMATCH_DTYPE, "matches")
        if out is None:
            out = np.zeros(max(len(mt), 1), dtype=DMATCH_DTYPE)
        if results is None:
            results = np.zeros(max(len(pairs), 1), dtype=RESULT_DTYPE)
        if (out.dtype != DMATCH_DTYPE or results.dtype != RESULT_DTYPE or len(out) < len(mt) or len(results) < len(pairs)
                or not out.flags["C_CONTIGUOUS"] or not results.flags["C_CONTIGUOUS"]):
            raise ValueError("out / results: contiguous DMATCH_DTYPE / RESULT_DTYPE arrays of at least len(matches) / len(pairs)")
        rc = self._lib.gms_filter_host_batch(self._h, kp.ctypes.data, frame_off.ctypes.data, wh.ctypes.data, n_frames,
                                             pairs.ctypes.data, len(pairs), mt.ctypes.data, int(bool(withRotation)),
                                             int(bool(withScale)), float(thresholdFactor), out.ctypes.data, results.ctypes.data)
        _check(rc, self._lib, "gms_filter_host_batch")
        return out[: len(mt)], results[: len(pairs)]

    # -- device-resident batch path (raw device pointers; torch tensors' data_ptr() are fine) ---------
    def reserve(self, n_pairs, max_m, withRotation=False, withScale=False):
        """gms_ctx_reserve: after it, filter_device calls of that shape neither allocate nor synchronise."""
        _check(self._lib.gms_ctx_reserve(self._h, int(n_pairs), int(max_m), int(bool(withRotation)), int(bool(withScale))),
               self._lib, "gms_ctx_reserve")

    def frame_table_bytes(self, total_kp):
        return int(self._lib.gms_frame_table_bytes(int(total_kp)))

    def normalize_device(self, d_kp, d_frame_off, d_wh, n_frames, total_kp, d_pts):
        _check(self._lib.gms_normalize_device(self._h, d_kp, d_frame_off, d_wh, int(n_frames), int(total_kp), d_pts),
               self._lib, "gms_normalize_device")

    def filter_device(self, d_pts, d_frame_off, n_frames, d_pairs, n_pairs, max_m, d_matches, d_out, d_results,
                      d_mask=None, withRotation=False, withScale=False, thresholdFactor=6.0):
        _check(self._lib.gms_filter_device(self._h, d_pts, d_frame_off, int(n_frames), d_pairs, int(n_pairs),
                                           int(max_m), d_matches, int(bool(withRotation)), int(bool(withScale)),
                                           float(thresholdFactor), d_out, d_results, d_mask or None),
               self._lib, "gms_filter_device")

    # -- brute-force descriptor matcher on the resident frame table (FeatureMatchUtil.cpp:66-68) -----------------------
    def bf_prepared_bytes(self, desc_kind, total_desc, n_frames):
        return int(self._lib.gms_bf_prepared_bytes(int(desc_kind), int(total_desc), int(n_frames)))

    def bf_prepare_device(self, desc_kind, d_desc, d_frame_off, n_frames, total_desc, d_prepared):
        _check(self._lib.gms_bf_prepare_device(self._h, int(desc_kind), d_desc, d_frame_off, int(n_frames), int(total_desc),
                                               d_prepared or None), self._lib, "gms_bf_prepare_device")

    def bfmatch_device(self, desc_kind, d_desc, d_prepared, total_desc, d_frame_off, n_frames, d_pairs, n_pairs, max_query,
                       d_matches):
        _check(self._lib.gms_bfmatch_device(self._h, int(desc_kind), d_desc, d_prepared or None, int(total_desc), d_frame_off,
                                            int(n_frames), d_pairs, int(n_pairs), int(max_query), d_matches),
               self._lib, "gms_bfmatch_device")

    # -- consumers of the filtered matches (DisparityUtil.cpp:179-201, SfMUtil.cpp:25-35) -------------------------------
    def disparity_device(self, d_kp1, n1, d_kp2, n2, d_matches, d_n_matches, max_matches, width, height, d_gt, disp_ratio,
                         d_disparity, d_work, d_stats):
        _check(self._lib.gms_disparity_device(self._h, d_kp1, int(n1), d_kp2, int(n2), d_matches, d_n_matches, int(max_matches),
                                              int(width), int(height), d_gt or None, int(disp_ratio), d_disparity, d_work,
                                              d_stats), self._lib, "gms_disparity_device")

    def gather_points_device(self, d_kp1, n1, d_kp2, n2, d_matches, d_n_matches, max_matches, d_coords1, d_coords2, d_status):
        _check(self._lib.gms_gather_points_device(self._h, d_kp1, int(n1), d_kp2, int(n2), d_matches, d_n_matches,
                                                  int(max_matches), d_coords1, d_coords2, d_status),
               self._lib, "gms_gather_points_device")

    def triangulate_device(self, camera, dist, P1, P2, d_coords1, d_coords2, d_n_matches, max_matches, d_points3d, d_stats):
        """gms_triangulate_device: camera = (fx, fy, cx, cy), dist = (k1, k2, p1, p2, k3) or None, P1 / P2 3 x 4 (host values)."""
        cam = np.ascontiguousarray(camera, dtype=np.float64).reshape(4)
        dc = None if dist is None else np.ascontiguousarray(dist, dtype=np.float64).reshape(5)
        p1 = np.ascontiguousarray(P1, dtype=np.float64).reshape(12)
        p2 = np.ascontiguousarray(P2, dtype=np.float64).reshape(12)
        _check(self._lib.gms_triangulate_device(self._h, cam.ctypes.data, None if dc is None else dc.ctypes.data, p1.ctypes.data,
                                                p2.ctypes.data, d_coords1, d_coords2, d_n_matches, int(max_matches), d_points3d,
                                                d_stats), self._lib, "gms_triangulate_device")

    def recover_pose_device(self, E, camera, d_coords1, d_coords2, d_n_matches, max_matches, d_in_mask, d_pose, d_out_mask):
        """gms_recover_pose_device: E 3 x 3 and camera = (fx, fy, cx, cy) are host values; d_pose receives a POSE_DTYPE record."""
        e = np.ascontiguousarray(E, dtype=np.float64).reshape(9)
        cam = np.ascontiguousarray(camera, dtype=np.float64).reshape(4)
        _check(self._lib.gms_recover_pose_device(self._h, e.ctypes.data, cam.ctypes.data, d_coords1, d_coords2, d_n_matches,
                                                 int(max_matches), d_in_mask, d_pose, d_out_mask), self._lib, "gms_recover_pose_device")

    # -- the same consumers for a whole batch (SfMUtil.cpp:25-82, DisparityUtil.cpp:170-201): one launch per stage ---------------------
    def gather_points_batch_device(self, d_kp, d_frame_off, n_frames, d_pairs, n_pairs, max_m, d_filtered, d_results, d_coords1,
                                   d_coords2, d_tv):
        _check(self._lib.gms_gather_points_batch_device(self._h, d_kp, d_frame_off, int(n_frames), d_pairs, int(n_pairs), int(max_m),
                                                        d_filtered, d_results, d_coords1, d_coords2, d_tv),
               self._lib, "gms_gather_points_batch_device")

    def find_essential_batch_device(self, camera, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_tv, prob=0.999, threshold=1.0,
                                    max_iters=1000):
        """cv::findEssentialMat(..., RANSAC, prob, threshold, mask) per pair; camera: a CAMERA_DTYPE record (types.make_camera).
        The defaults are OpenCV's (0.999, 1.0); the reference's call, SfMUtil.cpp:39, passes 0.7 -- pipeline.run_dataset defaults to that."""
        _check(self._lib.gms_find_essential_batch_device(self._h, camera.ctypes.data, float(prob), float(threshold), int(max_iters),
                                                         d_pairs, int(n_pairs), d_coords1, d_coords2, d_mask, d_tv),
               self._lib, "gms_find_essential_batch_device")

    def recover_pose_batch_device(self, camera, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_tv, use_in_mask=True):
        _check(self._lib.gms_recover_pose_batch_device(self._h, camera.ctypes.data, int(bool(use_in_mask)), d_pairs, int(n_pairs),
                                                       d_coords1, d_coords2, d_mask, d_tv), self._lib, "gms_recover_pose_batch_device")

    def triangulate_batch_device(self, camera, d_pairs, n_pairs, d_coords1, d_coords2, d_mask, d_points3d, d_tv):
        _check(self._lib.gms_triangulate_batch_device(self._h, camera.ctypes.data, d_pairs, int(n_pairs), d_coords1, d_coords2,
                                                      d_mask or None, d_points3d, d_tv), self._lib, "gms_triangulate_batch_device")

    def two_view_batch_device(self, camera, d_kp, d_frame_off, n_frames, d_pairs, n_pairs, max_m, d_filtered, d_results, d_coords1,
                              d_coords2, d_mask, d_points3d, d_tv, prob=0.999, threshold=1.0, max_iters=1000):
        """SfMUtil.cpp:25-82 for every pair of the batch: gather -> findEssentialMat -> recoverPose -> undistort + triangulate.
        prob defaults to OpenCV's 0.999; the reference passes 0.7 (SfMUtil.cpp:39): pass prob=0.7 to restate its flow."""
        _check(self._lib.gms_two_view_batch_device(self._h, camera.ctypes.data, float(prob), float(threshold), int(max_iters), d_kp,
                                                   d_frame_off, int(n_frames), d_pairs, int(n_pairs), int(max_m), d_filtered, d_results,
                                                   d_coords1, d_coords2, d_mask, d_points3d, d_tv), self._lib, "gms_two_view_batch_device")

    def disparity_batch_device(self, d_kp, d_frame_off, d_wh, n_frames, d_pairs, n_pairs, max_m, d_filtered, d_results, d_gt, gt_stride,
                               disp_ratio, d_disparity, map_stride, d_work, d_stats):
        _check(self._lib.gms_disparity_batch_device(self._h, d_kp, d_frame_off, d_wh, int(n_frames), d_pairs, int(n_pairs), int(max_m),
                                                    d_filtered, d_results, d_gt or None, int(gt_stride), int(disp_ratio), d_disparity,
                                                    int(map_stride), d_work, d_stats), self._lib, "gms_disparity_batch_device")

    def detect_workspace_bytes(self, width, height, n_images, max_keypoints):
        return int(self._lib.gms_detect_workspace_bytes(int(width), int(height), int(n_images), int(max_keypoints)))

    def detect_batch_device(self, d_images, n_images, width, height, threshold, max_keypoints, d_ws, ws_bytes, d_kp, d_desc, d_counts):
        _check(self._lib.gms_detect_batch_device(self._h, d_images, int(n_images), int(width), int(height), int(threshold), int(max_keypoints),
                                                 d_ws, int(ws_bytes), d_kp, d_desc, d_counts), self._lib, "gms_detect_batch_device")

    def describe_device(self, d_image, width, height, d_kp, n, d_ws, ws_bytes, d_desc, d_status):
        _check(self._lib.gms_describe_device(self._h, d_image, int(width), int(height), d_kp, int(n), d_ws, int(ws_bytes), d_desc, d_status),
               self._lib, "gms_describe_device")

    def selftest_five_point(self, x1, x2):
        """gms_selftest_five_point: x1, x2 [n_samples, 5, 2] normalised points -> list of [k, 3, 3] model arrays, one per sample."""
        x1 = np.asarray(x1, dtype=np.float64).reshape(-1, 5, 2)
        x2 = np.asarray(x2, dtype=np.float64).reshape(-1, 5, 2)
        n = len(x1)
        pts = np.ascontiguousarray(np.concatenate([x1[:, :, 0], x1[:, :, 1], x2[:, :, 0], x2[:, :, 1]], axis=1))
        models, counts = np.zeros((max(n, 1), 90)), np.zeros(max(n, 1), dtype=np.int32)
        _check(self._lib.gms_selftest_five_point(self._h, pts.ctypes.data, n, models.ctypes.data, counts.ctypes.data), self._lib, "selftest_five_point")
        return [models[i, :9 * counts[i]].reshape(-1, 3, 3) for i in range(n)]

    def selftest_threshold(self, T, n, score, factor):
        T = np.ascontiguousarray(T, dtype=np.int32)
        n = np.ascontiguousarray(n, dtype=np.int32)
        score = np.ascontiguousarray(score, dtype=np.int32)
        out = np.zeros(len(T), dtype=np.uint8)
        _check(self._lib.gms_selftest_threshold(self._h, T.ctypes.data, n.ctypes.data, score.ctypes.data,
                                                float(factor), len(T), out.ctypes.data), self._lib, "selftest")
        return out


_default_ctx = None


def matchGMS(size1, size2, keypoints1, keypoints2, matches1to2, withRotation=False, withScale=False,
             thresholdFactor=6.0):
    """Drop-in for cv::xfeatures2d::matchGMS; returns matchesGMS (the surviving DMatch, in input order).

    size = (width, height) like cv::Size. Raises GmsError instead of the reference's undefined behaviour
    on out-of-domain input."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = GmsContext(0)
    return _default_ctx.match(size1, size2, keypoints1, keypoints2, matches1to2, withRotation, withScale,
                              thresholdFactor)
