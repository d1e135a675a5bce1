"""CPU: pins for the oracle's pieces. The reference holds no tests or golden vectors for this path
(parity unpinned); what CAN be pinned is pinned here: the constant tables byte for byte
against the reference's DLL (when /root/reference is present), and the arithmetic of each piece against
hand-derived known answers taken from the DLL's disassembly."""
import math
import os
import struct

import numpy as np
import pytest

DLL = "/root/reference/SfM-GMS/bin/opencv_xfeatures2d452.dll"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.skipif(not os.path.exists(DLL), reason="reference DLL not present on this box")
def test_tables_match_reference_dll(oracle):
    import ctypes as C
    lib = oracle.load()
    data = open(DLL, "rb").read()
    # mRotationPatterns: .rdata VA 0x18012f520 -> file offset 0x12df20, 8 x 9 int32
    rot = struct.unpack("<72i", data[0x12DF20:0x12DF20 + 288])
    mine = (C.c_int * 72).in_dll(lib, "gms_ref_rotation_patterns")
    assert tuple(mine) == rot
    # mScaleRatios: .data VA 0x1802c5008 -> file offset 0x2c2c08; slots 2, 3 are filled by the static
    # initialiser (1/sqrt2, sqrt2), the others are literal
    sc = struct.unpack("<5d", data[0x2C2C08:0x2C2C08 + 40])
    assert sc[0] == 1.0 and sc[1] == 0.5 and sc[4] == 2.0 and sc[2] == 0.0 and sc[3] == 0.0
    assert [lib.gms_ref_scale_ratio(i) for i in range(5)] == [1.0, 0.5, 1.0 / math.sqrt(2.0), math.sqrt(2.0), 2.0]
    # the 0.5 added before the fp64 floor of the shifted grids: VA 0x18012dfc0 -> file offset 0x12c9c0
    assert struct.unpack("<d", data[0x12C9C0:0x12C9C8])[0] == 0.5


def test_rotation_patterns_are_rotations_of_the_ring(oracle):
    import ctypes as C
    pat = np.array((C.c_int * 72).in_dll(oracle.load(), "gms_ref_rotation_patterns")).reshape(8, 9)
    ring = [0, 1, 2, 5, 8, 7, 6, 3]  # the 8 neighbours clockwise, as positions in the 3x3 block
    for rot in range(8):
        assert pat[rot][4] == 5  # centre stays
        assert sorted(pat[rot]) == list(range(1, 10))
        # pattern rot is the identity ring rotated by rot steps
        assert [pat[rot][ring[(k + rot) % 8]] for k in range(8)] == [pat[0][ring[k]] for k in range(8)]


def test_rotation_patterns_arithmetic_form(oracle):
    """The closed form the HIP kernel uses instead of the table (gms_kernels.hip: rotated_position)."""
    import ctypes as C
    pat = np.array((C.c_int * 72).in_dll(oracle.load(), "gms_ref_rotation_patterns")).reshape(8, 9)
    ring_index = [0, 1, 2, 7, -1, 3, 6, 5, 4]
    for rot in range(8):
        for k in range(9):
            if k == 4:
                continue
            q = (0x36785210 >> ((((ring_index[k] - rot) & 7)) << 2)) & 15
            assert q == pat[rot][k] - 1
            assert ((0x24924 >> (q << 1)) & 3) - 1 == q % 3 - 1 and ((0x2a540 >> (q << 1)) & 3) - 1 == q // 3 - 1


def test_right_grids(oracle):
    import ctypes as C
    lib = oracle.load()
    got = []
    for s in range(5):
        w, h = C.c_int(), C.c_int()
        lib.gms_ref_right_grid(s, C.byref(w), C.byref(h))
        got.append((w.value, h.value))
    assert got == [(20, 20), (10, 10), (14, 14), (28, 28), (40, 40)]  # cvRound(20 * ratio), DLL@0x180048c10


def test_grid_index_left_known_answers(oracle):
    lib = oracle.load()
    L = lib.gms_ref_grid_index_left
    # (nx, ny) -> cell for types 1..4; 20*0.26 = 5.2, 20*0.53 = 10.6
    assert [L(0.26, 0.53, t) for t in (1, 2, 3, 4)] == [5 + 10 * 20, 5 + 10 * 20, 5 + 11 * 20, 5 + 11 * 20]
    # 20*0.275 = 5.5: the shifted x jumps to 6
    assert [L(0.275, 0.1, t) for t in (1, 2, 3, 4)] == [5 + 40, 6 + 40, 5 + 40, 6 + 40]
    # last half cell: 20*0.98 = 19.6 -> shifted x = 20 -> -1 (one common bounds test)
    assert [L(0.98, 0.5, t) for t in (1, 2, 3, 4)] == [19 + 200, -1, 19 + 200, -1]
    assert [L(0.5, 0.99, t) for t in (1, 2, 3, 4)] == [10 + 380, 10 + 380, -1, -1]
    # exact borders
    assert L(0.0, 0.0, 1) == 0 and L(0.0, 0.0, 4) == 0
    assert L(0.05, 0.05, 1) == 1 + 20  # 20 * fl32(0.05) = 1.0000000149 in fp32 -> rounds to 1.0
    # the product is rounded to fp32 BEFORE the widening + 0.5: nx just below 0.475 whose fp32 product is 9.5
    nx = np.nextafter(np.float32(0.475), np.float32(0))
    prod = np.float32(20) * nx
    want = int(math.floor(float(prod) + 0.5))
    assert L(float(nx), 0.0, 2) == want


def test_grid_index_right_has_no_bounds_test(oracle):
    R = oracle.load().gms_ref_grid_index_right
    assert R(0.26, 0.53, 20, 20) == 5 + 10 * 20
    assert R(0.26, 0.53, 14, 14) == 3 + 7 * 14
    assert R(0.999, 0.999, 40, 40) == 39 + 39 * 40
    assert R(1.0, 0.0, 20, 20) == 20  # x == W spills into the next row in the reference: no test there


def test_normalize_is_fp32_divide(oracle):
    N = oracle.load().gms_ref_normalize
    for v, e in [(1919.5, 1920), (0.1, 3), (1079.99, 1080), (123.456, 777)]:
        assert N(v, e) == float(np.float32(v) / np.float32(e))


def test_threshold_is_strict_greater_in_fp64(oracle):
    T = oracle.load().gms_ref_threshold_rejects
    assert T(36, 9, 12, 6.0) == 0  # 6 * sqrt(4) == 12: kept
    assert T(36, 9, 11, 6.0) == 1
    assert T(9, 9, 6, 6.0) == 0 and T(9, 9, 5, 6.0) == 1
    assert T(2, 9, 2, 6.0) == 1  # 6 * sqrt(2/9) = 2.83
    assert T(1, 4, 3, 6.0) == 0  # 6 * 0.5 == 3
    rng = np.random.default_rng(9)
    for _ in range(2000):
        t, n, s = int(rng.integers(0, 5000)), int(rng.integers(1, 10)), int(rng.integers(0, 200))
        assert T(t, n, s, 6.0) == int(math.sqrt(t / n) * 6.0 > s)


def test_cell_mapping_matches_the_reference_binary(oracle):
    """tests/golden/refdll_grid_index.npz holds what the reference's OWN DLL code returned for 6.7k points
    (GMSMatcher::getGridIndexLeft / getGridIndexRight executed out of opencv_xfeatures2d452.dll by
    tests/golden/refdll_runner.c). Both restatements must reproduce every integer."""
    import gms_ref_sparse
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdll_grid_index.npz"))
    nxy, left, right, dims = z["nxy"], z["left"], z["right"], z["right_dims"]
    assert len(nxy) > 6000 and left.min() == -1 and left.max() == 399
    lib = oracle.load()
    for i in range(len(nxy)):
        nx, ny = float(nxy[i, 0]), float(nxy[i, 1])
        assert [lib.gms_ref_grid_index_left(nx, ny, t) for t in (1, 2, 3, 4)] == left[i].tolist(), (i, nx, ny)
        assert [lib.gms_ref_grid_index_right(nx, ny, int(d), int(d)) for d in dims] == right[i].tolist(), (i, nx, ny)
    for t in (1, 2, 3, 4):
        assert np.array_equal(gms_ref_sparse._left_cells(nxy, t), left[:, t - 1])
    for k, d in enumerate(dims):
        assert np.array_equal(gms_ref_sparse._right_cells(nxy, int(d), int(d)), right[:, k])


def test_binning_matches_the_reference_binary(oracle):
    """tests/golden/refdll_assign_pairs.npz: GMSMatcher::assignMatchPairs executed out of the reference DLL for grid
    types 1..4 (driven as run() drives it) on three right grids -- the (left cell, right cell) it records per match,
    the per-cell counts and every non-zero of the motion matrix. The oracle's assign_match_pairs must agree."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdll_assign_pairs.npz"))
    for tag in ("g20", "g14", "g40"):
        wr, n1, n2, m = (int(v) for v in z[tag + "_dims"])
        rc, pairs, nleft, motion = oracle.assign_pairs(z[tag + "_p1"], z[tag + "_p2"], z[tag + "_matches"], wr, wr)
        assert rc == 0
        for t in range(4):
            assert np.array_equal(pairs[t], z[f"{tag}_pairs{t + 1}"]), (tag, t)
            assert np.array_equal(nleft[t], z[f"{tag}_nleft{t + 1}"]), (tag, t)
            l, r = np.nonzero(motion[t])
            got = np.stack([l, r, motion[t][l, r]], axis=1).astype(np.int32)
            assert np.array_equal(got, z[f"{tag}_motion{t + 1}"]), (tag, t)
        # the shifted grid types really do reject the last half cell, and type 1's right cell is reused
        assert (z[f"{tag}_pairs2"][:, 0] == -1).any() and (z[f"{tag}_pairs1"][:, 0] >= 0).all()
        assert np.array_equal(z[f"{tag}_pairs1"][:, 1], z[f"{tag}_pairs4"][:, 1])


def test_cell_verification_matches_the_reference_binary(oracle):
    """tests/golden/refdll_verify_cells.npz: the body of GMSMatcher::verifyCellPairs (arg-max scan, rotated 3 x 3
    neighbour sums, sqrt(T / n) * factor, the '>' test) executed out of the reference DLL for rotation types 1..8 on
    nine motion matrices over all five right-grid sizes and four threshold factors, one of them meeting the threshold
    with equality. The oracle's verify_cell_pairs must return the same mCellPairs for every left cell."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdll_verify_cells.npz"))
    tags = sorted(k[:-5] for k in z.files if k.endswith("_dims"))
    assert len(tags) == 9
    seen_grids, accepted, rejected = set(), 0, 0
    for tag in tags:
        wr, hr = (int(v) for v in z[tag + "_dims"])
        seen_grids.add(wr)
        motion = np.zeros((400, wr * hr), dtype=np.int32)
        nz = z[tag + "_motion"]
        motion[nz[:, 0], nz[:, 1]] = nz[:, 2]
        want = z[tag + "_cell_pairs"]
        for rot in range(1, 9):
            got = oracle.verify_cells(motion, z[tag + "_nleft"], wr, hr, rot, float(z[tag + "_factor"]))
            assert np.array_equal(got, want[rot - 1]), (tag, rot, np.nonzero(got != want[rot - 1])[0][:8])
        accepted += int((want >= 0).sum())
        rejected += int((want == -2).sum())
    assert seen_grids == {10, 14, 20, 28, 40} and accepted > 3000 and rejected > 10000
    eq = z["eq_cell_pairs"][0].reshape(20, 20)
    assert eq[5, 5] == 5 * 20 + 5          # 36 == 6 * sqrt(324 / 9): equality is not a rejection
    assert eq[10, 10] == -2 and eq[9, 11] == -2 and eq[12, 10] == 12 * 20 + 10   # one vote short around cell 210
    assert eq[0, 0] == -1 and eq[1, 1] == -2 and eq[0, 5] == -2                   # empty row; its neighbour; a border


def test_neighbour_tables_match_the_reference_binary(oracle):
    """tests/golden/refdll_nb9.npz: GMSMatcher::initalizeNeighbors -- and getNB9 through it -- executed out of the reference DLL
    (its operator new / delete pointed at the host allocator: refdll_runner.c "nb9") for the left grid, the five right grids of
    setScale and three odd grids. Both restatements must produce the same [w * h, 9] tables; the verify fixture above was
    generated with the DLL's tables, so nothing the pinned verifyCellPairs body consumed comes from a restatement any more."""
    import gms_ref_sparse
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdll_nb9.npz"))
    grids = [tuple(int(v) for v in g) for g in z["grids"]]
    assert {(20, 20), (10, 10), (14, 14), (28, 28), (40, 40)} <= set(grids) and len(grids) == 8
    for w, h in grids:
        want = z[f"nb9_{w}x{h}"]
        assert want.shape == (w * h, 9)
        assert np.array_equal(oracle.neighbors(w, h), want), (w, h)
        assert np.array_equal(np.array([gms_ref_sparse._neighbors(i, w, h) for i in range(w * h)], dtype=np.int32), want), (w, h)
    t = z["nb9_20x20"]
    assert list(t[0]) == [-1, -1, -1, -1, 0, 1, -1, 20, 21] and list(t[399]) == [378, 379, -1, 398, 399, -1, -1, -1, -1]
    assert list(z["nb9_7x3"][10]) == [2, 3, 4, 9, 10, 11, 16, 17, 18] and list(z["nb9_1x1"][0]) == [-1, -1, -1, -1, 0, -1, -1, -1, -1]


def test_normalize_points_matches_the_reference_binary(oracle):
    """tests/golden/refdll_normalize.npz: GMSMatcher::normalizePoints executed out of the reference DLL on cv::KeyPoint records of
    eight image sizes (its four-at-a-time loop and its tail). The oracle's divide must give the same bits."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdll_normalize.npz"))
    lib = oracle.load()
    n_cases = len([k for k in z.files if k.endswith("_size")])
    assert n_cases == 8
    for i in range(n_cases):
        w, h = (int(v) for v in z[f"c{i}_size"])
        xy, want = z[f"c{i}_xy"], z[f"c{i}_normalized"]
        got = np.array([[lib.gms_ref_normalize(float(x), w), lib.gms_ref_normalize(float(y), h)] for x, y in xy], dtype=np.float32)
        assert got.view(np.uint32).tobytes() == want.view(np.uint32).tobytes(), (w, h)
        assert (want[0] == 0).all() and want[-1, 0] < 1.0        # (0, 0) stays 0; the last representable x stays inside [0, 1)


def test_set_scale_matches_the_reference_binary(oracle):
    """tests/golden/refdll_setscale.npz: the DLL's static initialiser of mScaleRatios and the head of GMSMatcher::setScale, executed
    out of the reference DLL for scales 0..4 on the 20 x 20 left grid and on a 15 x 25 one (7.5 -> 8, 12.5 -> 12, 37.5 -> 38:
    cvRound is round-half-even). Right grid, cell count, and the n_right x 9 int32 neighbour table it then asks cv::Mat::zeros for."""
    import ctypes as C
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refdll_setscale.npz"))
    lib = oracle.load()
    assert [oracle.load().gms_ref_scale_ratio(s) for s in range(5)] == z["ratios"].tolist()
    for key, (lw, lh) in (("left20x20", (20, 20)), ("left15x25", (15, 25))):
        for s in range(5):
            wr, hr = C.c_int(0), C.c_int(0)
            lib.gms_ref_right_grid_from(lw, lh, s, C.byref(wr), C.byref(hr))
            assert [wr.value, hr.value, wr.value * hr.value, wr.value * hr.value, 9, 4] == z[key][s].tolist(), (key, s)
    assert z["left20x20"][:, 0].tolist() == [20, 10, 14, 28, 40] and z["left15x25"][1].tolist()[:2] == [8, 12]


def test_marking_loop_matches_the_reference_binary(oracle):
    """The tail of GMSMatcher::run executed out of the DLL (tests/golden/refdll_runner.c "mark": the marking loop of one grid type,
    the grid-type loop's exit, the count of the mask's bits, RVA 0x48acd-0x48bd4) for grid types 1..4 in sequence: the oracle's
    mark_inliers / count_mask -- the functions its run() is made of -- must leave the same mask and return the same count."""
    z = np.load(os.path.join(GOLDEN, "refdll_mark.npz"))
    tags = sorted({k.split("_")[0] for k in z.files})
    assert tags == list("abcdef")
    for tag in tags:
        m = int(z[tag + "_m"])
        mask = np.zeros(m, dtype=np.uint8)
        for t in range(1, 5):
            count = oracle.mark_inliers(z[f"{tag}_pairs{t}"].reshape(m, 2), z[f"{tag}_cell_pairs{t}"], mask)
            words = z[f"{tag}_mask_words{t}"]
            want = np.array([(int(words[i // 32]) >> (i % 32)) & 1 for i in range(m)], dtype=np.uint8)
            assert count == int(z[f"{tag}_count{t}"]) == int(want.sum()), (tag, t)
            assert np.array_equal(mask, want), (tag, t)


def test_hypothesis_selection_matches_the_reference_binary(oracle):
    """GMSMatcher::getInlierMask executed WHOLE out of the DLL with its calls of setScale / run re-pointed at a script player
    (refdll_runner.c "select"): the sequence of calls (scale outer, rotation inner), the strict '>' (ties keep the first), the
    mask that ends up in the caller's vector and the count returned. The oracle's select_hypothesis -- the loop its real path
    runs -- is driven by the same scripts."""
    z = np.load(os.path.join(GOLDEN, "refdll_select.npz"))
    names = sorted({k[:-len("_counts")] for k in z.files if k.endswith("_counts")})
    assert len(names) == 11
    for name in names:
        counts, masks = z[name + "_counts"], z[name + "_masks"]
        for rot in (0, 1):
            for scale in (0, 1):
                key = f"{name}_rot{rot}_scale{scale}"
                best, bs, br, mask, calls = oracle.select_hypothesis(rot, scale, counts, masks)
                assert best == int(z[key + "_ret"]), key
                assert calls.tolist() == z[key + "_calls"].tolist(), key
                want_calls = [v for s in range(5 if scale else 1) for v in [100 + s] + list(range(1, (8 if rot else 1) + 1))]
                assert calls.tolist() == want_calls, key
                if int(z[key + "_bits"]) == 0:       # the DLL never assigned the caller's vector: no hypothesis had an inlier
                    assert best == 0 and not mask.any() and (rot or scale), key
                else:
                    assert np.array_equal(mask, z[key + "_mask"]), key
                    if best > 0:
                        assert np.array_equal(mask, masks[bs, br - 1]) and counts[bs, br - 1] == best, key
