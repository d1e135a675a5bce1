"""The ingest format (SURVEY.md 8f "f2"; include/gms.h "ingest format"): a sequence's keypoints, descriptors, pairs and putative
matches in one little-endian file, the records verbatim cv::KeyPoint / cv::DMatch. The reference has no on-disk form
(FeatureMatchUtil.cpp:9-12,58-68 keep everything in std::vector / cv::Mat); a caller dumps its vectors with gms_dataset_write
(csrc/gms_io.cpp) or `save` here, and `load` hands back the arrays the batch API takes. numpy only: no GPU involved."""
import ctypes as C

import numpy as np

from .types import DMATCH_DTYPE, KEYPOINT_DTYPE, PAIR_DTYPE

MAGIC = b"GMSFRM01"
_HEADER = np.dtype([("magic", "S8"), ("n_frames", "<u4"), ("desc_kind", "<u4"), ("total_kp", "<u8"), ("n_pairs", "<u8"),
                    ("total_matches", "<u8")])
_ROW = {-1: (None, 0), 0: (np.uint8, 32), 1: (np.float32, 128)}


class Dataset:
    """frames: list of KEYPOINT_DTYPE arrays; sizes: [(w, h)]; descriptors: list of [n, 32] uint8 / [n, 128] float32 or None;
    pairs (PAIR_DTYPE) index matches (DMATCH_DTYPE) by match_off."""

    def __init__(self, frames, sizes, descriptors=None, desc_kind=-1, pairs=None, matches=None):
        self.frames, self.sizes, self.descriptors, self.desc_kind = list(frames), [tuple(s) for s in sizes], descriptors, int(desc_kind)
        self.pairs = np.zeros(0, dtype=PAIR_DTYPE) if pairs is None else np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        self.matches = np.zeros(0, dtype=DMATCH_DTYPE) if matches is None else np.ascontiguousarray(matches, dtype=DMATCH_DTYPE)


def save(path, ds):
    counts = np.array([len(f) for f in ds.frames], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(counts)]).astype("<i8")
    hdr = np.zeros(1, dtype=_HEADER)
    hdr["magic"], hdr["n_frames"], hdr["desc_kind"] = MAGIC, len(ds.frames), ds.desc_kind + 1
    hdr["total_kp"], hdr["n_pairs"], hdr["total_matches"] = int(off[-1]), len(ds.pairs), len(ds.matches)
    with open(path, "wb") as f:
        f.write(hdr.tobytes())
        f.write(np.asarray(ds.sizes, dtype="<i4").reshape(-1).tobytes())
        if len(ds.frames):
            f.write(off.tobytes())
        for fr in ds.frames:
            f.write(np.ascontiguousarray(fr, dtype=KEYPOINT_DTYPE).tobytes())
        dt, width = _ROW[ds.desc_kind]
        if dt is not None:
            for d in ds.descriptors:
                f.write(np.ascontiguousarray(d, dtype=dt).reshape(-1, width).tobytes())
        f.write(ds.pairs.tobytes())
        f.write(ds.matches.tobytes())


def load(path):
    import os
    with open(path, "rb") as f:
        hdr = np.frombuffer(f.read(_HEADER.itemsize), dtype=_HEADER)
        if len(hdr) != 1 or hdr["magic"][0] != MAGIC or int(hdr["desc_kind"][0]) > 2:
            raise ValueError(f"{path}: not a GMSFRM01 file")
        n, kind = int(hdr["n_frames"][0]), int(hdr["desc_kind"][0]) - 1
        total, n_pairs, total_m = int(hdr["total_kp"][0]), int(hdr["n_pairs"][0]), int(hdr["total_matches"][0])
        # the header must describe exactly the bytes that follow it (same rule as gms_dataset_read), before anything is read
        row = {-1: 0, 0: 32, 1: 512}[kind]
        payload = 8 * n + (8 * (n + 1) if n else 0) + (28 + row) * total + 24 * n_pairs + 16 * total_m
        if max(total, n_pairs, total_m) > 1 << 40 or os.fstat(f.fileno()).st_size != _HEADER.itemsize + payload:
            raise ValueError(f"{path}: header does not describe this file")
        take = lambda dt, count: np.frombuffer(f.read(np.dtype(dt).itemsize * count), dtype=dt, count=count)
        wh = take("<i4", 2 * n).reshape(-1, 2)
        off = take("<i8", n + 1) if n else np.zeros(1, dtype=np.int64)
        if int(off[-1]) != total or (np.diff(off) < 0).any():
            raise ValueError(f"{path}: frame offsets do not match the keypoint count")
        kp = take(KEYPOINT_DTYPE, total)
        dt, width = _ROW[kind]
        desc = take(dt, total * width).reshape(-1, width) if dt is not None else None
        pairs = take(PAIR_DTYPE, n_pairs)
        matches = take(DMATCH_DTYPE, total_m)
        if n_pairs and not ((pairs["frame_a"] >= 0) & (pairs["frame_a"] < n) & (pairs["frame_b"] >= 0) & (pairs["frame_b"] < n) &
                            (pairs["m"] >= 0) & (pairs["match_off"] >= 0) & (pairs["match_off"] + pairs["m"] <= total_m)).all():
            raise ValueError(f"{path}: a pair names a frame or a match range outside the file")
    frames = [kp[off[i]:off[i + 1]] for i in range(n)]
    descs = [desc[off[i]:off[i + 1]] for i in range(n)] if desc is not None else None
    return Dataset(frames, [tuple(x) for x in wh.tolist()], descs, kind, pairs, matches)


class _CDataset(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("desc_kind", C.c_int32), ("n_pairs", C.c_int64), ("total_matches", C.c_int64),
                ("wh", C.c_void_p), ("frame_off", C.c_void_p), ("keypoints", C.c_void_p), ("descriptors", C.c_void_p),
                ("pairs", C.c_void_p), ("matches", C.c_void_p), ("owner", C.c_void_p)]


def load_c(path):
    """The same file through the C reader of the library (gms_dataset_read, csrc/gms_io.cpp): what a C / C++ caller gets. Returns a
    Dataset whose arrays are copies (the C block is released before returning). Raises OSError on GMS_ERR_IO."""
    from .capi import load_library
    lib = load_library()
    cd = _CDataset()
    rc = lib.gms_dataset_read(str(path).encode(), C.byref(cd))
    if rc != 0:
        raise OSError(f"gms_dataset_read({path}): {lib.gms_error_string(rc).decode()}")
    try:
        n, kind = int(cd.n_frames), int(cd.desc_kind)

        def arr(ptr, dtype, count):
            if not ptr or count == 0:
                return np.zeros(0, dtype=dtype)
            buf = (C.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()
        wh = arr(cd.wh, "<i4", 2 * n).reshape(-1, 2)
        off = arr(cd.frame_off, "<i8", n + 1) if n else np.zeros(1, dtype=np.int64)
        total = int(off[-1])
        kp = arr(cd.keypoints, KEYPOINT_DTYPE, total)
        dt, width = _ROW[kind]
        desc = arr(cd.descriptors, dt, total * width).reshape(-1, width) if dt is not None else None
        pairs = arr(cd.pairs, PAIR_DTYPE, int(cd.n_pairs))
        matches = arr(cd.matches, DMATCH_DTYPE, int(cd.total_matches))
    finally:
        lib.gms_dataset_free(C.byref(cd))
    frames = [kp[off[i]:off[i + 1]] for i in range(n)]
    descs = [desc[off[i]:off[i + 1]] for i in range(n)] if desc is not None else None
    return Dataset(frames, [tuple(x) for x in wh.tolist()], descs, kind, pairs, matches)


def save_c(path, ds):
    """Dataset -> file through the C writer (gms_dataset_write)."""
    from .capi import load_library
    lib = load_library()
    counts = np.array([len(f) for f in ds.frames], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(counts)]).astype("<i8")
    wh = np.ascontiguousarray(np.asarray(ds.sizes, dtype="<i4").reshape(-1))
    kp = np.concatenate([np.ascontiguousarray(f, dtype=KEYPOINT_DTYPE) for f in ds.frames]) if len(ds.frames) else np.zeros(0, dtype=KEYPOINT_DTYPE)
    dt, width = _ROW[ds.desc_kind]
    desc = np.concatenate([np.ascontiguousarray(d, dtype=dt).reshape(-1, width) for d in ds.descriptors]) if dt is not None else None
    pairs, matches = np.ascontiguousarray(ds.pairs, dtype=PAIR_DTYPE), np.ascontiguousarray(ds.matches, dtype=DMATCH_DTYPE)
    cd = _CDataset(len(ds.frames), ds.desc_kind, len(pairs), len(matches), wh.ctypes.data, off.ctypes.data, kp.ctypes.data if len(kp) else None,
                   desc.ctypes.data if desc is not None and len(desc) else None, pairs.ctypes.data if len(pairs) else None,
                   matches.ctypes.data if len(matches) else None, None)
    rc = lib.gms_dataset_write(str(path).encode(), C.byref(cd))
    if rc != 0:
        raise OSError(f"gms_dataset_write({path}): {lib.gms_error_string(rc).decode()}")
