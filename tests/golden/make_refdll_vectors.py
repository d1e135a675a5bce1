#!/usr/bin/env python3
"""Generates tests/golden/refdll_*.npz (grid index, assign pairs, verify cells). refdll_grid_index.npz: (nx, ny) -> cell indices AS RETURNED BY THE REFERENCE'S OWN BINARY.

refdll_runner.c maps SfM-GMS/bin/opencv_xfeatures2d452.dll and calls its GMSMatcher::getGridIndexLeft /
getGridIndexRight (two leaf functions) on the inputs built here. Needs /root/reference; run in the build container:
    python tests/golden/make_refdll_vectors.py
The fixture pins the float -> cell mapping (the bit-exactness-critical arithmetic) of oracle/gms_ref.c to the reference."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DLL = "/root/reference/SfM-GMS/bin/opencv_xfeatures2d452.dll"


def inputs():
    rng = np.random.default_rng(0x5F3759DF)
    f32 = np.float32
    pts = [np.stack([rng.uniform(0, 1, 3000), rng.uniform(0, 1, 3000)], axis=1)]
    # pixel coordinates divided by real image extents (what normalizePoints produces)
    for w, h in ((1920, 1080), (640, 480), (3840, 2160), (450, 375), (2016, 1512), (1390, 1110)):
        x = rng.uniform(0, w, 400).astype(f32)
        y = rng.uniform(0, h, 400).astype(f32)
        x[:40] = np.rint(x[:40])  # integer pixels
        pts.append(np.stack([x / f32(w), y / f32(h)], axis=1))
        # pixels whose products land on / next to cell and half-cell borders
        k = np.arange(0, 41, dtype=np.float64)
        bx = (k * w / 40.0).astype(f32)
        by = (k * h / 40.0).astype(f32)
        for d in (-1, 0, 1):
            xx = bx.copy()
            yy = by.copy()
            for _ in range(abs(d)):
                xx = np.nextafter(xx, f32(np.inf) if d > 0 else f32(-np.inf))
                yy = np.nextafter(yy, f32(np.inf) if d > 0 else f32(-np.inf))
            keep = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            pts.append(np.stack([xx[keep] / f32(w), yy[keep] / f32(h)], axis=1))
    # normalised values right at k/20, k/40 and their neighbours in fp32, all four combinations with a random partner
    k = np.arange(0, 40)
    edge = np.concatenate([(k / 40.0).astype(f32)] + [np.nextafter((k / 40.0).astype(f32), f32(s)) for s in (0, 1)]
                          + [np.nextafter(np.nextafter((k / 40.0).astype(f32), f32(s)), f32(s)) for s in (0, 1)])
    edge = edge[(edge >= 0) & (edge < 1)]
    other = rng.uniform(0, 1, len(edge)).astype(f32)
    pts += [np.stack([edge, other], axis=1), np.stack([other, edge], axis=1), np.stack([edge, edge[::-1]], axis=1)]
    pts.append(np.array([[0, 0], [0.99999994, 0.99999994], [1e-30, 1e-30], [0.5, 0.5], [0.975, 0.975], [0.97500002, 0.025]]))
    a = np.concatenate([p.astype(f32) for p in pts])
    a = a[(a[:, 0] >= 0) & (a[:, 0] < 1) & (a[:, 1] >= 0) & (a[:, 1] < 1)]
    return np.ascontiguousarray(a, dtype=f32)


def assign_fixture():
    """GMSMatcher::assignMatchPairs run out of the DLL for grid types 1..4 on two right grids: the pairs it records,
    the per-cell counts and the motion matrix (kept as its non-zeros)."""
    rng = np.random.default_rng(0x5F3759DF ^ 2)
    out = {}
    for tag, (wr, n1, n2, m) in {"g20": (20, 1200, 1100, 1500), "g14": (14, 900, 900, 1000), "g40": (40, 700, 800, 900)}.items():
        p1 = np.stack([rng.uniform(0, 1, n1), rng.uniform(0, 1, n1)], axis=1).astype(np.float32)
        p1[: n1 // 4] = (np.round(p1[: n1 // 4] * 40) / 40).astype(np.float32).clip(0, 0.999)  # on cell / half-cell borders
        p1[n1 // 4: n1 // 3, 0] = rng.uniform(0.95, 1, n1 // 3 - n1 // 4).astype(np.float32)   # the last half cell
        p2 = np.stack([rng.uniform(0, 1, n2), rng.uniform(0, 1, n2)], axis=1).astype(np.float32)
        p2 = np.minimum(p2, np.float32(0.99999))
        mt = np.stack([rng.integers(0, n1, m), rng.integers(0, n2, m)], axis=1).astype(np.int32)
        mt[: m // 2, 1] = np.minimum(mt[: m // 2, 0], n2 - 1)  # repeated partners
        with tempfile.TemporaryDirectory() as tmp:
            exe = os.path.join(tmp, "refdll_runner")
            subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "refdll_runner.c")])
            fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            with open(fin, "wb") as f:
                f.write(np.array([wr, wr, n1, n2, m], dtype=np.int32).tobytes())
                f.write(p1.tobytes()); f.write(p2.tobytes()); f.write(mt.tobytes())
            subprocess.check_call([exe, DLL, fin, fout, "assign"])
            raw = np.fromfile(fout, dtype=np.int32)
        per = 2 * m + 400 + 400 * wr * wr
        assert len(raw) == 4 * per
        out[tag + "_dims"] = np.array([wr, n1, n2, m], dtype=np.int32)
        out[tag + "_p1"], out[tag + "_p2"], out[tag + "_matches"] = p1, p2, mt
        for t in range(4):
            blk = raw[t * per:(t + 1) * per]
            out[f"{tag}_pairs{t + 1}"] = blk[: 2 * m].reshape(m, 2).copy()
            out[f"{tag}_nleft{t + 1}"] = blk[2 * m: 2 * m + 400].copy()
            motion = blk[2 * m + 400:].reshape(400, wr * wr)
            l, r = np.nonzero(motion)
            out[f"{tag}_motion{t + 1}"] = np.stack([l, r, motion[l, r]], axis=1).astype(np.int32)
            assert motion.sum() == blk[2 * m: 2 * m + 400].sum()
    np.savez_compressed(os.path.join(HERE, "refdll_assign_pairs.npz"), **out)
    print("assign fixture", os.path.getsize(os.path.join(HERE, "refdll_assign_pairs.npz")), "bytes")


def verify_motion(rng, wr, quarter_turns, lam, noise):
    """A motion matrix with structure: most left cells vote for the cell a rotated, shrunk copy of the grid puts them
    in (Poisson counts around the threshold), some with a second right cell tied for the maximum, plus scattered votes."""
    nr = wr * wr
    mo = np.zeros((400, nr), dtype=np.int32)
    a = quarter_turns * np.pi / 4
    for i in range(400):
        cx, cy = (i % 20 + 0.5) / 20 - 0.5, (i // 20 + 0.5) / 20 - 0.5
        rx = (cx * np.cos(a) - cy * np.sin(a)) * 0.7 + 0.5
        ry = (cx * np.sin(a) + cy * np.cos(a)) * 0.7 + 0.5
        j = int(rx * wr) + int(ry * wr) * wr
        if rng.random() < 0.85:
            mo[i, j] += rng.poisson(lam)
            if rng.random() < 0.3:
                mo[i, rng.integers(0, nr)] = mo[i, j]
        for _ in range(rng.poisson(noise)):
            mo[i, rng.integers(0, nr)] += 1
    return mo


def verify_fixture():
    """The body of GMSMatcher::verifyCellPairs run out of the DLL (refdll_runner.c "verify") for rotation types 1..8:
    mCellPairs for every left cell of several motion matrices / right grids / threshold factors."""
    rng = np.random.default_rng(0x5F3759DF ^ 3)
    cases = {}
    for tag, (wr, turns, lam, noise, factor) in {
            "a": (20, 0, 4, 1, 6.0), "b": (20, 2, 5, 2, 6.0), "c": (10, 1, 6, 1, 6.0), "d": (14, 3, 3, 2, 4.0),
            "e": (28, 5, 5, 1, 6.0), "f": (40, 7, 4, 1, 2.5), "g": (20, 4, 30, 3, 6.0), "h": (40, 6, 9, 0, 0.1)}.items():
        mo = verify_motion(rng, wr, turns, lam, noise)
        cases[tag] = (wr, factor, mo, mo.sum(axis=1).astype(np.int32))
    # threshold met with equality: 36 matches per cell, 4 of them on the diagonal -> interior score 36 = 6 * sqrt(36);
    # one vote fewer in a neighbour tips its whole 3 x 3 surroundings
    mo = np.zeros((400, 400), dtype=np.int32)
    mo[np.arange(400), np.arange(400)] = 4
    mo[210, 210] = 3
    mo[0, 0] = 0  # an empty row next to the corner
    cases["eq"] = (20, 6.0, mo, np.full(400, 36, dtype=np.int32))
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "refdll_runner")
        subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "refdll_runner.c")])
        for tag, (wr, factor, mo, nleft) in cases.items():
            fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            with open(fin, "wb") as f:
                f.write(np.array([wr, wr], dtype=np.int32).tobytes())
                f.write(np.float64(factor).tobytes())
                f.write(nleft.tobytes()); f.write(mo.tobytes())
            subprocess.check_call([exe, DLL, fin, fout, "verify"])
            cp = np.fromfile(fout, dtype=np.int32).reshape(8, 400)
            l, r = np.nonzero(mo)
            out[tag + "_dims"] = np.array([wr, wr], dtype=np.int32)
            out[tag + "_factor"] = np.float64(factor)
            out[tag + "_nleft"] = nleft
            out[tag + "_motion"] = np.stack([l, r, mo[l, r]], axis=1).astype(np.int32)
            out[tag + "_cell_pairs"] = cp
            print("verify", tag, "accepted per rotation", (cp >= 0).sum(axis=1).tolist(), "empty", int((cp[0] == -1).sum()))
    np.savez_compressed(os.path.join(HERE, "refdll_verify_cells.npz"), **out)
    print("verify fixture", os.path.getsize(os.path.join(HERE, "refdll_verify_cells.npz")), "bytes")


def nb9_fixture():
    """GMSMatcher::initalizeNeighbors (and getNB9 through it) run out of the DLL for the left grid, the five right grids of
    setScale and two non-square grids: the [w * h, 9] neighbour tables as the DLL fills them."""
    grids = [(20, 20), (10, 10), (14, 14), (28, 28), (40, 40), (7, 3), (1, 5), (1, 1)]
    out = {"grids": np.array(grids, dtype=np.int32)}
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "refdll_runner")
        subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "refdll_runner.c")])
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.int32(len(grids)).tobytes())
            f.write(np.array(grids, dtype=np.int32).tobytes())
        subprocess.check_call([exe, DLL, fin, fout, "nb9"])
        raw = np.fromfile(fout, dtype=np.int32)
    pos = 0
    for w, h in grids:
        out[f"nb9_{w}x{h}"] = raw[pos:pos + 9 * w * h].reshape(w * h, 9).copy()
        pos += 9 * w * h
    assert pos == len(raw)
    np.savez_compressed(os.path.join(HERE, "refdll_nb9.npz"), **out)
    print("nb9 fixture", os.path.getsize(os.path.join(HERE, "refdll_nb9.npz")), "bytes")


def normalize_fixture():
    """GMSMatcher::normalizePoints run out of the DLL on cv::KeyPoint records of several image sizes (the reference's own images:
    450 x 375, 1920 x 1080, 2016 x 1512, 1390 x 1110, 2594 x 1131, and BASELINE's 640 x 480 / 3840 x 2160), counts that exercise
    both its four-at-a-time loop and its tail, coordinates that are integers, sub-pixel, tiny, on the far border."""
    rng = np.random.default_rng(0x5F3759DF ^ 4)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = {}
    cases = [(450, 375, 301), (1920, 1080, 1000), (2016, 1512, 7), (1390, 1110, 3), (2594, 1131, 258), (640, 480, 1), (3840, 2160, 513), (3, 7, 5)]
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "refdll_runner")
        subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "refdll_runner.c")])
        for i, (w, h, n) in enumerate(cases):
            kp = np.zeros(n, dtype=kp_dtype)
            kp["x"] = rng.uniform(0, w, n).astype(np.float32)
            kp["y"] = rng.uniform(0, h, n).astype(np.float32)
            kp["x"][: n // 3] = np.floor(kp["x"][: n // 3])                       # integer pixels (the per-pixel grids of DisparityUtil.cpp:123-133)
            kp["y"][n // 5: n // 2] = np.floor(kp["y"][n // 5: n // 2])
            kp["x"][-1], kp["y"][-1] = np.nextafter(np.float32(w), np.float32(0)), np.float32(1e-30)
            kp["x"][0], kp["y"][0] = 0.0, 0.0
            kp["size"], kp["angle"], kp["response"], kp["octave"], kp["class_id"] = 31.0, -1.0, rng.uniform(0, 1, n), 3, -1
            fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            with open(fin, "wb") as f:
                f.write(np.array([n, w, h], dtype=np.int32).tobytes())
                f.write(kp.tobytes())
            subprocess.check_call([exe, DLL, fin, fout, "normalize"])
            res = np.fromfile(fout, dtype=np.float32).reshape(n, 2)
            out[f"c{i}_size"] = np.array([w, h], dtype=np.int32)
            out[f"c{i}_xy"] = np.stack([kp["x"], kp["y"]], axis=1)
            out[f"c{i}_normalized"] = res
    np.savez_compressed(os.path.join(HERE, "refdll_normalize.npz"), **out)
    print("normalize fixture", os.path.getsize(os.path.join(HERE, "refdll_normalize.npz")), "bytes")


def setscale_fixture():
    """The head of GMSMatcher::setScale run out of the DLL for scales 0..4 (after the DLL's own static initialiser has filled the
    two dynamic entries of mScaleRatios): the ratios, the right grid per scale, and the shape of the neighbour table it asks for."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "refdll_runner")
        subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "refdll_runner.c")])
        for lw, lh in ((20, 20), (15, 25)):
            fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            with open(fin, "wb") as f:
                f.write(np.array([lw, lh], dtype=np.int32).tobytes())
            subprocess.check_call([exe, DLL, fin, fout, "setscale"])
            raw = open(fout, "rb").read()
            out["ratios"] = np.frombuffer(raw[:40], dtype=np.float64).copy()
            out[f"left{lw}x{lh}"] = np.frombuffer(raw[40:], dtype=np.int32).reshape(5, 6).copy()
    np.savez_compressed(os.path.join(HERE, "refdll_setscale.npz"), **out)
    print("setscale fixture", out["ratios"].tolist(), out["left20x20"].tolist())


def _runner(tmp):
    exe = os.path.join(tmp, "refdll_runner")
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "refdll_runner.c")])
    return exe


def mark_fixture():
    """The tail of GMSMatcher::run (marking loop of one grid type, the count of the mask, RVA 0x48acd on) run out of the DLL for grid
    types 1..4 in sequence on hand-laid-out mvMatchPairs / mCellPairs: the mask after every type and the count the code returns."""
    rng = np.random.default_rng(0x5F3759DF ^ 5)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = _runner(tmp)
        for tag, (m, nr) in {"a": (1000, 400), "b": (77, 100), "c": (4097, 1600), "d": (32, 196), "e": (1, 784), "f": (0, 400)}.items():
            pairs, cps = [], []
            for t in range(4):
                cp = rng.integers(0, nr, 400).astype(np.int32)
                cp[rng.random(400) < 0.25] = -1          # empty cells
                cp[rng.random(400) < 0.25] = -2          # rejected cells
                first = rng.integers(-1, 400, m).astype(np.int32)           # -1: the left point is on no cell of this grid type
                second = rng.integers(0, nr, m).astype(np.int32)
                hit = rng.random(m) < 0.4                                    # matches whose right cell IS their cell's partner
                second[hit] = np.where(first[hit] >= 0, cp[np.maximum(first[hit], 0)], second[hit])
                second[rng.random(m) < 0.02] = -1                            # a right cell of -1 equals an "empty" cell pair: the DLL's compare decides
                second[rng.random(m) < 0.02] = -2
                pairs.append(np.stack([first, second], axis=1).astype(np.int32))
                cps.append(cp)
            fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            with open(fin, "wb") as f:
                f.write(np.int32(m).tobytes())
                for t in range(4):
                    f.write(pairs[t].tobytes()); f.write(cps[t].tobytes())
            subprocess.check_call([exe, DLL, fin, fout, "mark"])
            raw = np.fromfile(fout, dtype=np.uint32)
            words = (m + 31) // 32
            assert len(raw) == 4 * (words + 1)
            out[tag + "_m"] = np.int32(m)
            for t in range(4):
                blk = raw[t * (words + 1):(t + 1) * (words + 1)]
                out[f"{tag}_pairs{t + 1}"], out[f"{tag}_cell_pairs{t + 1}"] = pairs[t], cps[t]
                out[f"{tag}_mask_words{t + 1}"] = blk[:words].copy()
                out[f"{tag}_count{t + 1}"] = np.int32(blk[words])
            print("mark", tag, "counts", [int(out[f"{tag}_count{t + 1}"]) for t in range(4)])
    np.savez_compressed(os.path.join(HERE, "refdll_mark.npz"), **out)
    print("mark fixture", os.path.getsize(os.path.join(HERE, "refdll_mark.npz")), "bytes")


def select_fixture():
    """GMSMatcher::getInlierMask itself run out of the DLL on scripted setScale / run (refdll_runner.c "select"): for every flag
    combination and several scripts of counts -- distinct, tied, all zero, the maximum first / last, falling, rising -- the count it
    returns, the mask it leaves in the caller's vector and the sequence of calls it makes."""
    rng = np.random.default_rng(0x5F3759DF ^ 6)
    out = {}
    scripts = {}
    m = 70
    base = rng.integers(1, 60, (5, 8)).astype(np.int32)
    scripts["random"] = base
    t = base.copy(); t[:] = 17
    scripts["all_tied"] = t
    scripts["all_zero"] = np.zeros((5, 8), dtype=np.int32)
    t = base.copy(); t[0, 0] = 99
    scripts["first_is_max"] = t
    t = base.copy(); t[4, 7] = 99
    scripts["last_is_max"] = t
    t = base.copy(); t[2, 3] = 80; t[3, 1] = 80; t[1, 6] = 80
    scripts["ties_across_scales"] = t
    t = base.copy(); t[1, 2] = 80; t[1, 5] = 80
    scripts["ties_inside_a_scale"] = t
    scripts["rising"] = np.arange(40, dtype=np.int32).reshape(5, 8)
    scripts["falling"] = (40 - np.arange(40, dtype=np.int32)).reshape(5, 8)
    t = np.zeros((5, 8), dtype=np.int32); t[3, 4] = 1
    scripts["one_inlier_late"] = t
    t = np.zeros((5, 8), dtype=np.int32); t[0, 1:] = 5; t[1:, 0] = 5
    scripts["first_run_empty"] = t
    words = (m + 31) // 32
    with tempfile.TemporaryDirectory() as tmp:
        exe = _runner(tmp)
        for name, counts in scripts.items():
            masks = rng.integers(0, 2, (5, 8, m)).astype(np.uint8)       # a distinct mask per hypothesis (the counts are the script's, not popcounts)
            packed = np.zeros((5, 8, words), dtype=np.uint32)
            for i in range(m):
                packed[:, :, i // 32] |= masks[:, :, i].astype(np.uint32) << np.uint32(i % 32)
            out[f"{name}_counts"], out[f"{name}_masks"] = counts, masks
            for rot in (0, 1):
                for scale in (0, 1):
                    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
                    with open(fin, "wb") as f:
                        f.write(np.array([m, rot, scale], dtype=np.int32).tobytes())
                        f.write(counts.tobytes()); f.write(packed.tobytes())
                    subprocess.check_call([exe, DLL, fin, fout, "select"])
                    raw = open(fout, "rb").read()
                    ret = np.frombuffer(raw[:4], dtype=np.int32)[0]
                    bits = np.frombuffer(raw[4:12], dtype=np.int64)[0]
                    w = np.frombuffer(raw[12:12 + 4 * words], dtype=np.uint32)
                    n_log = np.frombuffer(raw[12 + 4 * words:16 + 4 * words], dtype=np.int32)[0]
                    log = np.frombuffer(raw[16 + 4 * words:], dtype=np.int32)[:n_log]
                    got = np.array([(w[i // 32] >> np.uint32(i % 32)) & 1 for i in range(m)], dtype=np.uint8)
                    key = f"{name}_rot{rot}_scale{scale}"
                    out[key + "_ret"], out[key + "_bits"], out[key + "_mask"], out[key + "_calls"] = np.int32(ret), np.int64(bits), got, log.copy()
            print("select", name, {f"r{r}s{sc}": int(out[f"{name}_rot{r}_scale{sc}_ret"]) for r in (0, 1) for sc in (0, 1)})
    np.savez_compressed(os.path.join(HERE, "refdll_select.npz"), **out)
    print("select fixture", os.path.getsize(os.path.join(HERE, "refdll_select.npz")), "bytes")


def chain_fixture():
    """matchGMS's whole computation as a chain of the DLL's own pieces (refdll_runner.c "chain": normalizePoints, getInlierMask with
    its setScale / run calls re-pointed at drivers that run the DLL's setScale head, initalizeNeighbors, assignMatchPairs, the body of
    verifyCellPairs and the marking / counting tail of run) on the inputs of every golden case of this directory, all four flag
    combinations: the mask the DLL's getInlierMask leaves and the count it returns."""
    import glob
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = _runner(tmp)
        for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
            name = os.path.basename(path)[:-4]
            z = np.load(path)
            if "xy1" not in z.files or "query" not in z.files:
                continue
            xy1, xy2, q, t = z["xy1"], z["xy2"], z["query"], z["train"]
            m = len(q)
            kps = []
            for xy in (xy1, xy2):
                kp = np.zeros(len(xy), dtype=kp_dtype)
                kp["x"], kp["y"], kp["size"], kp["angle"], kp["class_id"] = xy[:, 0], xy[:, 1], 31.0, -1.0, -1
                kps.append(kp)
            for rot in (0, 1):
                for scale in (0, 1):
                    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
                    with open(fin, "wb") as f:
                        f.write(np.array([len(xy1), len(xy2), m, z["size1"][0], z["size1"][1], z["size2"][0], z["size2"][1], rot, scale], dtype=np.int32).tobytes())
                        f.write(np.float64(6.0).tobytes())
                        f.write(kps[0].tobytes()); f.write(kps[1].tobytes())
                        f.write(np.stack([q, t], axis=1).astype(np.int32).tobytes())
                    subprocess.check_call([exe, DLL, fin, fout, "chain"])
                    raw = open(fout, "rb").read()
                    words = (m + 31) // 32
                    ret = int(np.frombuffer(raw[:4], dtype=np.int32)[0])
                    bits = int(np.frombuffer(raw[4:12], dtype=np.int64)[0])
                    w = np.frombuffer(raw[12:12 + 4 * words], dtype=np.uint32)
                    mask = np.unpackbits(w.view(np.uint8), bitorder="little")[:m] if m else np.zeros(0, dtype=np.uint8)
                    assert bits in (0, m) and (bits == m or not mask.any())
                    key = f"{name}_r{rot}s{scale}"
                    out[key + "_mask"] = np.packbits(mask, bitorder="little")
                    out[key + "_ret"] = np.int32(ret)
                    same = (f"mask_r{rot}s{scale}" in z.files and np.array_equal(np.unpackbits(z[f"mask_r{rot}s{scale}"], bitorder="little")[:m], mask))
                    print("chain", key, "count", ret, "kept", int(mask.sum()), "== golden" if same else "!= GOLDEN")
    np.savez_compressed(os.path.join(HERE, "refdll_chain.npz"), **out)
    print("chain fixture", os.path.getsize(os.path.join(HERE, "refdll_chain.npz")), "bytes")


def main():
    if not os.path.exists(DLL):
        sys.exit("reference DLL not present: this generator runs only where /root/reference is mounted")
    pts = inputs()
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "refdll_runner")
        subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "refdll_runner.c")])
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.int32(len(pts)).tobytes())
            f.write(pts.tobytes())
        subprocess.check_call([exe, DLL, fin, fout])
        res = np.fromfile(fout, dtype=np.int32).reshape(-1, 9)
    assert len(res) == len(pts)
    np.savez_compressed(os.path.join(HERE, "refdll_grid_index.npz"), nxy=pts, left=res[:, :4], right=res[:, 4:],
                        right_dims=np.array([20, 10, 14, 28, 40], dtype=np.int32))
    assign_fixture()
    verify_fixture()
    nb9_fixture()
    normalize_fixture()
    setscale_fixture()
    mark_fixture()
    select_fixture()
    chain_fixture()
    print(len(pts), "points;", "left range", res[:, :4].min(), res[:, :4].max(), "; file",
          os.path.getsize(os.path.join(HERE, "refdll_grid_index.npz")), "bytes")


if __name__ == "__main__":
    main()
