"""CPU: the ingest format -- the C writer/reader in libgms_hip.so (csrc/gms_io.cpp) and the numpy mirror (sfm-gms_amd/io.py)
agree byte for byte, records are verbatim cv::KeyPoint / cv::DMatch, broken files are refused."""
import ctypes as C
import importlib

import numpy as np
import pytest


def _dataset(pkg, synth, kind):
    io = importlib.import_module("sfm-gms_amd.io")
    size = (640, 480)
    frames = synth.make_sequence(3, 3, size=size, n_kp=70) + [synth.make_keypoints(np.zeros((0, 2), dtype=np.float32))]
    descs = None
    if kind >= 0:
        descs = synth.sequence_descriptors(3, 3, 70, "orb" if kind == 0 else "sift") + [np.zeros((0, 32 if kind == 0 else 128))]
    pairs = np.zeros(2, dtype=pkg.PAIR_DTYPE)
    m0, m1 = synth.sequence_matches(1, 70, 70), synth.sequence_matches(2, 70, 70)[:33]
    pairs[0], pairs[1] = (0, 1, 70, 0, 0), (1, 2, 33, 0, 70)
    return io, io.Dataset(frames, [size, size, (800, 600), (1, 1)], descs, kind, pairs, np.concatenate([m0, m1]))


@pytest.mark.parametrize("kind", [-1, 0, 1])
def test_c_and_numpy_agree(pkg, synth, tmp_path, kind):
    io, ds = _dataset(pkg, synth, kind)
    lib = pkg.load_library()
    p_np, p_c = str(tmp_path / "np.gmsf"), str(tmp_path / "c.gmsf")
    io.save(p_np, ds)
    cd = io._CDataset()
    assert lib.gms_dataset_read(p_np.encode(), C.byref(cd)) == 0      # the C reader takes what numpy wrote ...
    assert (cd.n_frames, cd.desc_kind, cd.n_pairs, cd.total_matches) == (4, kind, 2, 103)
    assert (cd.descriptors is None) == (kind < 0)
    assert lib.gms_dataset_write(p_c.encode(), C.byref(cd)) == 0      # ... and writes the same bytes back
    lib.gms_dataset_free(C.byref(cd))
    assert cd.owner is None and open(p_np, "rb").read() == open(p_c, "rb").read()
    back = io.load(p_c)
    assert back.sizes == ds.sizes and back.desc_kind == kind and back.pairs.tobytes() == ds.pairs.tobytes()
    assert back.matches.tobytes() == ds.matches.tobytes()
    assert all(a.tobytes() == np.ascontiguousarray(b, dtype=pkg.KEYPOINT_DTYPE).tobytes() for a, b in zip(back.frames, ds.frames))
    if kind >= 0:
        dt = np.uint8 if kind == 0 else np.float32
        assert all(a.tobytes() == np.ascontiguousarray(b, dtype=dt).tobytes() for a, b in zip(back.descriptors, ds.descriptors))


def test_broken_files_are_refused(pkg, synth, tmp_path):
    io, ds = _dataset(pkg, synth, 0)
    lib = pkg.load_library()
    good = str(tmp_path / "good.gmsf")
    io.save(good, ds)
    raw = open(good, "rb").read()
    cd = io._CDataset()
    for name, data in (("short", raw[:-5]), ("magic", b"X" + raw[1:]), ("empty", b"")):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        assert lib.gms_dataset_read(p.encode(), C.byref(cd)) == -7 and cd.owner is None
        with pytest.raises(ValueError):
            io.load(p)
    assert lib.gms_dataset_read(str(tmp_path / "missing").encode(), C.byref(cd)) == -7
    assert lib.gms_error_string(-7).decode().startswith("dataset file")


def test_crafted_headers_and_out_of_range_pairs_are_refused(pkg, synth, tmp_path):
    """A header whose counts wrap the byte sizes (n_pairs = 2^60, total_matches = 2^59: 24 * 2^60 + 16 * 2^59 = 2^65 wraps to a tiny
    allocation) or simply exceed the file, and pairs that name frames / match ranges outside the file: GMS_ERR_IO, nothing allocated."""
    io, ds = _dataset(pkg, synth, -1)
    lib = pkg.load_library()
    good = str(tmp_path / "good.gmsf")
    io.save(good, ds)
    raw = bytearray(open(good, "rb").read())
    hdr = np.frombuffer(bytes(raw[:40]), dtype=io._HEADER).copy()
    cd = io._CDataset()

    def refused(name, data):
        p = str(tmp_path / name)
        open(p, "wb").write(bytes(data))
        assert lib.gms_dataset_read(p.encode(), C.byref(cd)) == -7 and cd.owner is None
        with pytest.raises(ValueError):
            io.load(p)

    h = hdr.copy()
    h["n_frames"], h["total_kp"], h["n_pairs"], h["total_matches"] = 0, 0, 1 << 60, 1 << 59
    refused("wrap", h.tobytes() + bytes(raw[40:]))
    h = hdr.copy()
    h["total_matches"] = int(hdr["total_matches"][0]) + 1            # one match more than the file holds
    refused("long", h.tobytes() + bytes(raw[40:]))
    refused("padded", bytes(raw) + b"\0" * 16)                       # bytes the header does not account for
    pair_off = 40 + 8 * 4 + 8 * 5 + 28 * 210                         # header, wh, frame_off, keypoints (no descriptors)
    pairs = np.frombuffer(bytes(raw[pair_off:pair_off + 48]), dtype=pkg.PAIR_DTYPE).copy()
    assert pairs.tobytes() == ds.pairs.tobytes()
    for field, value in (("frame_b", 4), ("frame_a", -1), ("m", -3), ("match_off", 71), ("match_off", -1)):
        bad = pairs.copy()
        bad[field][1] = value
        refused(f"pair_{field}_{value}", bytes(raw[:pair_off]) + bad.tobytes() + bytes(raw[pair_off + 48:]))
