"""oracle/gms_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes loader for oracle/libgms_oracle.so (the C restatement of the reference's matchGMS; parity pinned
only in part -- see gms_ref.c). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
PAIR_DTYPE = np.dtype([("frame_a", "<i4"), ("frame_b", "<i4"), ("m", "<i4"), ("reserved", "<i4"),
                       ("match_off", "<i8")])
RESULT_DTYPE = np.dtype([("n_inliers", "<i4"), ("best_scale", "<i4"), ("best_rot", "<i4"), ("status", "<i4")])


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(_HERE, "libgms_oracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
    lib.gms_ref_match.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, i32, i32, dbl, vp,
                                  C.POINTER(i32), vp, vp]
    lib.gms_ref_match.restype = i32
    lib.gms_ref_batch.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, i32, dbl, vp, vp, vp, i32]
    lib.gms_ref_batch.restype = i32
    lib.gms_ref_grid_index_left.argtypes = [C.c_float, C.c_float, i32]
    lib.gms_ref_grid_index_left.restype = i32
    lib.gms_ref_grid_index_right.argtypes = [C.c_float, C.c_float, i32, i32]
    lib.gms_ref_grid_index_right.restype = i32
    lib.gms_ref_right_grid.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
    lib.gms_ref_right_grid.restype = None
    lib.gms_ref_right_grid_from.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.gms_ref_right_grid_from.restype = None
    lib.gms_ref_normalize.argtypes = [C.c_float, i32]
    lib.gms_ref_normalize.restype = C.c_float
    lib.gms_ref_threshold_rejects.argtypes = [i32, i32, i32, dbl]
    lib.gms_ref_threshold_rejects.restype = i32
    lib.gms_ref_assign_pairs.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.gms_ref_assign_pairs.restype = i32
    lib.gms_ref_verify_cells.argtypes = [vp, vp, i32, i32, i32, dbl, vp]
    lib.gms_ref_verify_cells.restype = i32
    lib.gms_ref_selftest_mark.argtypes = [vp, vp, i32, vp]
    lib.gms_ref_selftest_mark.restype = i32
    lib.gms_ref_selftest_select.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.gms_ref_selftest_select.restype = i32
    lib.gms_ref_neighbors.argtypes = [i32, i32, vp]
    lib.gms_ref_neighbors.restype = None
    lib.gms_ref_scale_ratio.argtypes = [i32]
    lib.gms_ref_scale_ratio.restype = dbl
    lib.disp_ref_map_and_rms.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp]
    lib.disp_ref_map_and_rms.restype = i32
    lib.sfm_ref_gather.argtypes = [vp, i32, vp, i32, vp, i32, vp, vp]
    lib.sfm_ref_gather.restype = i32
    lib.bf_ref_hamming256.argtypes = [vp, i32, vp, i32, vp]
    lib.bf_ref_hamming256.restype = i32
    lib.bf_ref_l2.argtypes = [vp, i32, vp, i32, i32, vp]
    lib.bf_ref_l2.restype = i32
    lib.det_ref_detect.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.det_ref_detect.restype = i32
    lib.det_ref_describe.argtypes = [vp, i32, i32, vp, i32, vp]
    lib.det_ref_describe.restype = i32
    lib.det_ref_maps.argtypes = [vp, i32, i32, vp, vp]
    lib.det_ref_maps.restype = None
    lib.det_ref_pattern.argtypes = [vp]
    lib.det_ref_pattern.restype = None
    _lib = lib
    return lib


def bf_match(query, train, hamming):
    """The reference's BFMatcher::match restated (oracle/bf_ref.c): uint8 [n, 32] rows under NORM_HAMMING, or float32
    [n, dim] rows under NORM_L2; one DMatch per query row, first minimum."""
    lib = load()
    out = np.zeros(max(len(query), 1), dtype=DMATCH_DTYPE)
    if hamming:
        q = np.ascontiguousarray(query, dtype=np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(train, dtype=np.uint8).reshape(-1, 32)
        rc = lib.bf_ref_hamming256(q.ctypes.data, len(q), t.ctypes.data, len(t), out.ctypes.data)
    else:
        q = np.ascontiguousarray(query, dtype=np.float32)
        t = np.ascontiguousarray(train, dtype=np.float32)
        rc = lib.bf_ref_l2(q.ctypes.data, len(q), t.ctypes.data, len(t), q.shape[1], out.ctypes.data)
    assert rc == 0
    return out[: len(q)]


def detect(image, threshold=20, max_keypoints=10000):
    """oracle/detect_ref.c: the build's own FAST-9 + steered-BRIEF keypoint source on an 8-bit grey image -> (keypoints, [n, 32] rows)."""
    lib = load()
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = img.shape
    kp = np.zeros(max(max_keypoints, 1), dtype=KEYPOINT_DTYPE)
    desc = np.zeros((max(max_keypoints, 1), 32), dtype=np.uint8)
    n = lib.det_ref_detect(img.ctypes.data, w, h, int(threshold), int(max_keypoints), kp.ctypes.data, desc.ctypes.data)
    return kp[:n].copy(), desc[:n].copy()


def describe(image, keypoints):
    """oracle/detect_ref.c: directions and descriptors at given integer keypoints (Feature2D::compute). Returns (rc, keypoints, rows)."""
    lib = load()
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = img.shape
    kp = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE).copy()
    desc = np.zeros((max(len(kp), 1), 32), dtype=np.uint8)
    rc = lib.det_ref_describe(img.ctypes.data, w, h, kp.ctypes.data, len(kp), desc.ctypes.data)
    return rc, kp, desc[: len(kp)]


def detect_maps(image):
    """(FAST score image, 5 x 5 box sums) of oracle/detect_ref.c."""
    lib = load()
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = img.shape
    score = np.zeros((h, w), dtype=np.uint8)
    box = np.zeros((h, w), dtype=np.uint16)
    lib.det_ref_maps(img.ctypes.data, w, h, score.ctypes.data, box.ctypes.data)
    return score, box


def detect_pattern():
    lib = load()
    pat = np.zeros((256, 4), dtype=np.int8)
    lib.det_ref_pattern(pat.ctypes.data)
    return pat


def match(size1, size2, kp1, kp2, matches, with_rotation=False, with_scale=False, threshold_factor=6.0):
    """Returns (rc, out, mask, result)."""
    lib = load()
    kp1 = np.ascontiguousarray(kp1, dtype=KEYPOINT_DTYPE)
    kp2 = np.ascontiguousarray(kp2, dtype=KEYPOINT_DTYPE)
    mt = np.ascontiguousarray(matches, dtype=DMATCH_DTYPE)
    out = np.zeros(max(len(mt), 1), dtype=DMATCH_DTYPE)
    mask = np.zeros(max(len(mt), 1), dtype=np.uint8)
    res = np.zeros(1, dtype=RESULT_DTYPE)
    n_out = C.c_int(0)
    rc = lib.gms_ref_match(kp1.ctypes.data, len(kp1), int(size1[0]), int(size1[1]),
                           kp2.ctypes.data, len(kp2), int(size2[0]), int(size2[1]),
                           mt.ctypes.data, len(mt), int(bool(with_rotation)), int(bool(with_scale)),
                           float(threshold_factor), out.ctypes.data, C.byref(n_out), mask.ctypes.data,
                           res.ctypes.data)
    return rc, out[: n_out.value].copy(), mask[: len(mt)].copy(), res[0]


def batch(kp_all, frame_off, wh, pairs, matches, with_rotation=False, with_scale=False, threshold_factor=6.0,
          n_threads=1, bufs=None):
    """Returns (n_failed, out, results, mask) over a batch of pairs. bufs = (out, mask, res) arrays to write into (timing loops pass
    the same, already touched, arrays every time so that no page fault of a fresh allocation lands inside the timed call)."""
    lib = load()
    kp_all = np.ascontiguousarray(kp_all, dtype=KEYPOINT_DTYPE)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
    wh = np.ascontiguousarray(wh, dtype=np.int32)
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    mt = np.ascontiguousarray(matches, dtype=DMATCH_DTYPE)
    if bufs is not None:
        out, mask, res = bufs
        assert len(out) >= max(len(mt), 1) and len(mask) >= max(len(mt), 1) and len(res) >= max(len(pairs), 1)
    else:
        out = np.zeros(max(len(mt), 1), dtype=DMATCH_DTYPE)
        mask = np.zeros(max(len(mt), 1), dtype=np.uint8)
        res = np.zeros(max(len(pairs), 1), dtype=RESULT_DTYPE)
    failed = lib.gms_ref_batch(kp_all.ctypes.data, frame_off.ctypes.data, wh.ctypes.data, len(frame_off) - 1,
                               pairs.ctypes.data, len(pairs), mt.ctypes.data, int(bool(with_rotation)),
                               int(bool(with_scale)), float(threshold_factor), out.ctypes.data,
                               res.ctypes.data, mask.ctypes.data, int(n_threads))
    return failed, out[: len(mt)], res[: len(pairs)], mask[: len(mt)]


def assign_pairs(p1, p2, matches, wr, hr):
    """assignMatchPairs for grid types 1..4 on normalised points: (rc, pairs[4][m][2], nleft[4][400], motion[4][400][wr*hr])."""
    lib = load()
    p1 = np.ascontiguousarray(p1, dtype=np.float32)
    p2 = np.ascontiguousarray(p2, dtype=np.float32)
    mt = np.ascontiguousarray(matches, dtype=np.int32)
    m = len(mt)
    pairs = np.zeros((4, m, 2), dtype=np.int32)
    nleft = np.zeros((4, 400), dtype=np.int32)
    motion = np.zeros((4, 400, wr * hr), dtype=np.int32)
    rc = lib.gms_ref_assign_pairs(p1.ctypes.data, p2.ctypes.data, mt.ctypes.data, m, int(wr), int(hr),
                                  pairs.ctypes.data, nleft.ctypes.data, motion.ctypes.data)
    return rc, pairs, nleft, motion


def neighbors(gw, gh):
    """initalizeNeighbors / getNB9 for a gw x gh grid: int32 [gw * gh, 9]."""
    lib = load()
    out = np.zeros((gw * gh, 9), dtype=np.int32)
    lib.gms_ref_neighbors(int(gw), int(gh), out.ctypes.data)
    return out


def verify_cells(motion, nleft, wr, hr, rotation_type, factor=6.0):
    """verifyCellPairs on a dense [400, wr*hr] motion matrix and [400] per-cell counts -> int32[400] cell pairs."""
    lib = load()
    motion = np.ascontiguousarray(motion, dtype=np.int32)
    nleft = np.ascontiguousarray(nleft, dtype=np.int32)
    assert motion.size == 400 * wr * hr and nleft.size == 400
    out = np.empty(400, dtype=np.int32)
    rc = lib.gms_ref_verify_cells(motion.ctypes.data, nleft.ctypes.data, int(wr), int(hr), int(rotation_type),
                                  float(factor), out.ctypes.data)
    assert rc == 0
    return out


def mark_inliers(pairs, cell_pairs, mask):
    """run()'s marking loop for one grid type on (pairs [m, 2], cell_pairs [400]); `mask` (uint8 [m]) accumulates in place.
    Returns the count run() would return after this grid type."""
    lib = load()
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    cp = np.ascontiguousarray(cell_pairs, dtype=np.int32)
    assert cp.size == 400 and mask.dtype == np.uint8 and mask.flags["C_CONTIGUOUS"] and len(mask) == len(pairs)
    return int(lib.gms_ref_selftest_mark(pairs.ctypes.data, cp.ctypes.data, len(pairs), mask.ctypes.data))


def select_hypothesis(with_rotation, with_scale, counts, masks):
    """getInlierMask's loop nest on scripted run() results: counts [5, 8], masks uint8 [5, 8, m] ->
    (best count, best scale, best rot, best mask uint8 [m], call log)."""
    lib = load()
    counts = np.ascontiguousarray(counts, dtype=np.int32).reshape(5, 8)
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    m = masks.shape[2]
    assert masks.shape == (5, 8, m)
    best_mask = np.zeros(max(m, 1), dtype=np.uint8)
    best = np.zeros(3, dtype=np.int32)
    log = np.zeros(96, dtype=np.int32)
    n_log = C.c_int(0)
    rc = lib.gms_ref_selftest_select(int(bool(with_rotation)), int(bool(with_scale)), m, counts.ctypes.data, masks.ctypes.data,
                                     best_mask.ctypes.data, best.ctypes.data, log.ctypes.data, C.byref(n_log))
    assert rc == 0
    return int(best[0]), int(best[1]), int(best[2]), best_mask[:m], log[:n_log.value].copy()


def disparity(kp1, kp2, matches, width, height, gt, disp_ratio):
    """DisparityUtil.cpp:179-201 restated (oracle/consumer_ref.c): (rc, map [h, w] uint8, count, sum_sq, max_abs, rms)."""
    lib = load()
    kp1 = np.ascontiguousarray(kp1, dtype=KEYPOINT_DTYPE)
    kp2 = np.ascontiguousarray(kp2, dtype=KEYPOINT_DTYPE)
    mt = np.ascontiguousarray(matches, dtype=DMATCH_DTYPE)
    out = np.zeros((height, width), dtype=np.uint8)
    cnt, ssq, mx, rms = C.c_int64(0), C.c_int64(0), C.c_int32(0), C.c_double(0)
    g = None if gt is None else np.ascontiguousarray(gt, dtype=np.uint8)
    rc = lib.disp_ref_map_and_rms(kp1.ctypes.data, len(kp1), kp2.ctypes.data, len(kp2), mt.ctypes.data, len(mt), int(width),
                                  int(height), None if g is None else g.ctypes.data, int(disp_ratio), out.ctypes.data,
                                  C.byref(cnt), C.byref(ssq), C.byref(mx), C.byref(rms))
    return rc, out, cnt.value, ssq.value, mx.value, rms.value


def gather(kp1, kp2, matches):
    """SfMUtil.cpp:25-35 restated: (rc, coords1 [m, 2], coords2 [m, 2]) float32."""
    lib = load()
    kp1 = np.ascontiguousarray(kp1, dtype=KEYPOINT_DTYPE)
    kp2 = np.ascontiguousarray(kp2, dtype=KEYPOINT_DTYPE)
    mt = np.ascontiguousarray(matches, dtype=DMATCH_DTYPE)
    c1 = np.zeros((max(len(mt), 1), 2), dtype=np.float32)
    c2 = np.zeros((max(len(mt), 1), 2), dtype=np.float32)
    rc = lib.sfm_ref_gather(kp1.ctypes.data, len(kp1), kp2.ctypes.data, len(kp2), mt.ctypes.data, len(mt), c1.ctypes.data,
                            c2.ctypes.data)
    return rc, c1[: len(mt)], c2[: len(mt)]
