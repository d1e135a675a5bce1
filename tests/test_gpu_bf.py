"""-m gpu: the brute-force descriptor matcher (gms_bfmatch_device) against oracle/bf_ref.c -- index-exact and distance-exact
(Hamming: integers; L2: SIFT's integer-valued rows make every fp32 sum exact, everything else runs the reference's own loop)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(ctx, pkg, synth, descs, kind, size=(1280, 720)):
    batch = importlib.import_module("sfm-gms_amd.batch")
    rng = np.random.default_rng(11)
    frames = [synth.make_keypoints(np.stack([rng.uniform(0, size[0] - 1, len(d)), rng.uniform(0, size[1] - 1, len(d))], axis=1))
              for d in descs]
    table = batch.FrameTable(ctx, frames, [size] * len(frames))
    return batch, table, batch.DescriptorTable(ctx, table, descs, kind)


def _pairs(pkg, counts, ab):
    pairs = np.zeros(len(ab), dtype=pkg.PAIR_DTYPE)
    off = 0
    for i, (a, b) in enumerate(ab):
        pairs[i] = (a, b, counts[a], 0, off)
        off += counts[a]
    return pairs


def _check(oracle, descs, pairs, got, hamming):
    for p in pairs:
        want = oracle.bf_match(descs[p["frame_a"]], descs[p["frame_b"]], hamming)
        o = int(p["match_off"])
        assert got[o:o + len(want)].tobytes() == want.tobytes(), (int(p["frame_a"]), int(p["frame_b"]))


def test_hamming_small_frames_every_tail(ctx, pkg, oracle, synth):
    rng = np.random.default_rng(3)
    counts = [1, 2, 63, 64, 65, 255, 256, 257, 511, 513, 700, 1025]     # around the wave's 128 columns and the workgroup's 512 queries
    descs = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for n in counts]
    descs[10][100:140] = descs[10][20:60]        # duplicate train rows: the lower index wins
    descs[11][:40] = descs[10][20:60]            # ... and queries that hit them exactly
    batch, table, dt = _tables(ctx, pkg, synth, descs, pkg.GMS_DESC_HAMMING256)
    ab = [(a, b) for a in range(len(counts)) for b in range(len(counts)) if a != b]
    pairs = _pairs(pkg, counts, ab)
    for use_prepared in (True, False):       # the matrix-core kernel, and the vector-ALU kernel on the raw rows
        got = batch.match_pairs(ctx, dt, pairs, use_prepared)
        _check(oracle, descs, pairs, got, True)
        assert (got["imgIdx"] == 0).all()


def test_hamming_many_pairs_take_the_four_rows_per_lane_kernel(ctx, pkg, oracle, synth):
    descs = synth.sequence_descriptors(21, 12, 1500, "orb")
    batch, table, dt = _tables(ctx, pkg, synth, descs, pkg.GMS_DESC_HAMMING256)
    ab = [(a, b) for a in range(12) for b in range(12) if a != b] * 5      # 660 pairs x 2 tiles >= 1024 blocks
    pairs = _pairs(pkg, [1500] * 12, ab)
    got = batch.match_pairs(ctx, dt, pairs, use_prepared=False)
    _check(oracle, descs, pairs[::37], got, True)
    assert got.tobytes() == batch.match_pairs(ctx, dt, pairs).tobytes()   # the matrix-core kernel agrees on every pair
    first = got[:1500]
    assert (first["trainIdx"] == first["queryIdx"]).mean() > 0.95   # the same scene point is the nearest neighbour


def test_hamming_10k_config2(ctx, pkg, oracle, synth):
    descs = synth.sequence_descriptors(22, 2, 10000, "orb")
    batch, table, dt = _tables(ctx, pkg, synth, descs, pkg.GMS_DESC_HAMMING256, size=(1920, 1080))
    pairs = _pairs(pkg, [10000, 10000], [(0, 1), (1, 0)])
    for use_prepared in (True, False):
        got = batch.match_pairs(ctx, dt, pairs, use_prepared)
        _check(oracle, descs, pairs, got, True)


def test_hamming_frames_beyond_one_row_chunk(ctx, pkg, oracle, synth):
    """The matrix-core kernel carries (train row mod 32768) in the accumulator's fraction and settles the running minimum once per
    32768 rows: equal distances in two chunks keep the lower row, a strictly smaller one in a later chunk wins, and the row right at
    the boundary is found."""
    rng = np.random.default_rng(8)
    n_t, n_q = 70000, 300
    train = rng.integers(0, 256, (n_t, 32), dtype=np.uint8)
    query = rng.integers(0, 256, (n_q, 32), dtype=np.uint8)
    for q in range(0, 60):         # the same row in chunk 0 and chunk 1 (and chunk 2): the first wins
        train[100 + q] = train[40000 + q] = train[66000 + q] = query[q]
    for q in range(60, 120):       # one bit off in chunk 0, exact in chunk 2
        train[200 + q] = query[q]
        train[200 + q, 0] ^= 1
        train[66000 + q] = query[q]
    for q in range(120, 180):      # exact around the boundaries themselves
        train[32768 - 30 + (q - 120)] = query[q]
    for q in range(180, 240):
        train[65536 - 30 + (q - 180)] = query[q]
    descs = [query, train]
    batch, table, dt = _tables(ctx, pkg, synth, descs, pkg.GMS_DESC_HAMMING256, size=(3840, 2160))
    pairs = _pairs(pkg, [n_q, n_t], [(0, 1)])
    got = batch.match_pairs(ctx, dt, pairs)
    _check(oracle, descs, pairs, got, True)
    want = np.concatenate([100 + np.arange(60), 66000 + np.arange(60, 120), 32768 - 30 + np.arange(60), 65536 - 30 + np.arange(60)])
    assert (got["trainIdx"][:240] == want).all()
    assert (got["distance"][:240] == 0).all()


def test_l2_sift_like_rows_on_the_matrix_cores(ctx, pkg, oracle, synth):
    rng = np.random.default_rng(4)
    counts = [1, 31, 64, 65, 257, 1000, 3000, 777, 513]
    descs = [np.clip(np.rint(rng.gamma(1.2, 22.0, (n, 128))), 0, 255).astype(np.float32) for n in counts]
    descs[7] = rng.integers(0, 256, (777, 128)).astype(np.float32)   # every value of the int8 operands' range, both signs
    descs[5][500:520] = descs[5][100:120]        # duplicate train rows
    descs[6][:20] = descs[5][100:120]
    descs[4][:] = 255.0                          # the largest norms and distances that can occur
    descs[3][:] = 0.0
    batch, table, dt = _tables(ctx, pkg, synth, descs, pkg.GMS_DESC_L2_F32X128)
    ab = [(a, b) for a in range(len(counts)) for b in range(len(counts)) if a != b]
    pairs = _pairs(pkg, counts, ab)
    got = batch.match_pairs(ctx, dt, pairs)
    _check(oracle, descs, pairs, got, False)


def test_l2_distances_one_apart_in_different_tiles(ctx, pkg, oracle, synth):
    """The matrix-core pass ranks train rows by floor((d^2 - c(query)) / 2): a later tile holding a distance ONE below an earlier
    tile's minimum can share that value, and must still win (and lose when it is one above)."""
    rng = np.random.default_rng(6)
    n_q, n_t = 96, 1500
    base = rng.integers(40, 200, (n_q, 128)).astype(np.float32)
    train = rng.integers(0, 256, (n_t, 128)).astype(np.float32)     # far from every query
    for q in range(n_q):
        k, first_is_closer = 1 + q % 5, (q // 5) % 2 == 0
        near, far = base[q].copy(), base[q].copy()
        near[rng.choice(128, k, replace=False)] += 1.0               # d^2 = k
        far[rng.choice(128, k + 1, replace=False)] -= 1.0            # d^2 = k + 1
        r1, r2 = 3 + 7 * q, 700 + 8 * q                              # tiles 0..10 and 10..22
        train[r1], train[r2] = (near, far) if first_is_closer else (far, near)
    descs = [base, train]
    batch, table, dt = _tables(ctx, pkg, synth, descs, pkg.GMS_DESC_L2_F32X128)
    pairs = _pairs(pkg, [n_q, n_t], [(0, 1)])
    got = batch.match_pairs(ctx, dt, pairs)
    _check(oracle, descs, pairs, got, False)
    want_rows = np.array([(3 + 7 * q) if (q // 5) % 2 == 0 else (700 + 8 * q) for q in range(n_q)])
    assert (got["trainIdx"][:n_q] == want_rows).all()


def test_l2_general_floats_take_the_reference_loop(ctx, pkg, oracle, synth):
    rng = np.random.default_rng(5)
    sift = synth.sequence_descriptors(23, 2, 800, "sift")
    descs = [sift[0], sift[1], rng.normal(0, 1, (500, 128)).astype(np.float32), (sift[1] * 0.5).astype(np.float32),
             rng.uniform(-3, 300, (257, 128)).astype(np.float32)]
    counts = [len(d) for d in descs]
    batch, table, dt = _tables(ctx, pkg, synth, descs, pkg.GMS_DESC_L2_F32X128)
    ab = [(a, b) for a in range(5) for b in range(5) if a != b]      # mixed: exact pairs and loop pairs in one launch
    pairs = _pairs(pkg, counts, ab)
    got = batch.match_pairs(ctx, dt, pairs)
    _check(oracle, descs, pairs, got, False)


def test_l2_10k_sift_config2(ctx, pkg, oracle, synth):
    descs = synth.sequence_descriptors(24, 2, 10000, "sift")
    batch, table, dt = _tables(ctx, pkg, synth, descs, pkg.GMS_DESC_L2_F32X128, size=(1920, 1080))
    pairs = _pairs(pkg, [10000, 10000], [(0, 1), (1, 0)])
    got = batch.match_pairs(ctx, dt, pairs)
    _check(oracle, descs, pairs, got, False)
    assert (got[:10000]["trainIdx"] == np.arange(10000)).mean() > 0.95


@pytest.mark.parametrize("kind", ["orb", "sift"])
@pytest.mark.parametrize("rot,scale", [(False, False), (True, True)])
def test_descriptors_to_filtered_matches(ctx, pkg, oracle, synth, kind, rot, scale):
    """The pipeline of FeatureMatchUtil.cpp:58-69 on the device: BFMatcher::match -> matchGMS, the match array never leaving HBM."""
    import torch
    batch = importlib.import_module("sfm-gms_amd.batch")
    size, n_frames, n_kp = (1280, 720), 5, 2500
    frames = synth.make_sequence(31, n_frames, size=size, n_kp=n_kp)
    descs = synth.sequence_descriptors(31, n_frames, n_kp, kind, outlier_frac=0.4)
    table = batch.FrameTable(ctx, frames, [size] * n_frames)
    k = pkg.GMS_DESC_HAMMING256 if kind == "orb" else pkg.GMS_DESC_L2_F32X128
    dt = batch.DescriptorTable(ctx, table, descs, k)
    ab = [(a, b) for a in range(n_frames) for b in range(a + 1, n_frames)]
    pairs = _pairs(pkg, [n_kp] * n_frames, ab)
    dev = table.device
    d_pairs = batch._to_dev(pairs, dev)
    total = len(ab) * n_kp
    d_matches = torch.zeros(total * 16, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(total * 16, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(len(ab) * 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    dt.match_device(d_pairs.data_ptr(), len(ab), n_kp, d_matches.data_ptr())
    ctx.filter_device(table.d_pts.data_ptr(), table.d_frame_off.data_ptr(), n_frames, d_pairs.data_ptr(), len(ab), n_kp,
                      d_matches.data_ptr(), d_out.data_ptr(), d_res.data_ptr(), None, rot, scale, 6.0)
    ctx.synchronize()
    out = d_out.cpu().numpy().view(pkg.DMATCH_DTYPE)
    res = d_res.cpu().numpy().view(pkg.RESULT_DTYPE)
    matches = np.concatenate([oracle.bf_match(descs[a], descs[b], kind == "orb") for a, b in ab])
    assert d_matches.cpu().numpy().view(pkg.DMATCH_DTYPE).tobytes() == matches.tobytes()
    kp_all = np.concatenate(frames)
    wh = np.array([size] * n_frames, dtype=np.int32).reshape(-1)
    failed, wout, wres, _ = oracle.batch(kp_all, table.frame_off_host, wh, pairs, matches, rot, scale, 6.0, 4)
    assert failed == 0 and res.tobytes() == wres.tobytes() and (res["n_inliers"] > 300).all()
    for i in range(len(pairs)):
        o, kk = int(pairs["match_off"][i]), int(res["n_inliers"][i])
        assert out[o:o + kk].tobytes() == wout[o:o + kk].tobytes()
