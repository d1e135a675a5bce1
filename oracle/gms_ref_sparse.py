"""oracle/gms_ref_sparse.py -- TEST INFRASTRUCTURE, NOT PRODUCT (parity pinned only in part, see gms_ref.c).

A second, independently structured restatement of the reference's cv::xfeatures2d::matchGMS
(SfM-GMS/bin/opencv_xfeatures2d452.dll, GMSMatcher, DLL@0x180046900..0x1800491ec), used only to
cross-check oracle/gms_ref.c: where the C oracle keeps the reference's dense 400 x N_right matrix and
loop order, this one keeps a dictionary of (left cell, right cell) -> count and vectorises the cell
mapping with numpy float32/float64 arithmetic. The two share no code.
"""
from collections import Counter
import math

import numpy as np

ROTATION_PATTERNS = (  # DLL .rdata 0x18012f520
    (1, 2, 3, 4, 5, 6, 7, 8, 9), (4, 1, 2, 7, 5, 3, 8, 9, 6), (7, 4, 1, 8, 5, 2, 9, 6, 3),
    (8, 7, 4, 9, 5, 1, 6, 3, 2), (9, 8, 7, 6, 5, 4, 3, 2, 1), (6, 9, 8, 3, 5, 7, 2, 1, 4),
    (3, 6, 9, 2, 5, 8, 1, 4, 7), (2, 3, 6, 1, 5, 9, 4, 7, 8))
SCALE_RATIOS = (1.0, 0.5, 1.0 / math.sqrt(2.0), math.sqrt(2.0), 2.0)  # DLL .data 0x1802c5008
LEFT = 20


def _round_half_even(v):  # cvRound = cvtsd2si
    return int(np.rint(v))


def _neighbors(idx, gw, gh):  # getNB9, DLL@0x180048030
    ix, iy = idx % gw, idx // gw
    nb = [-1] * 9
    for yi in (-1, 0, 1):
        for xi in (-1, 0, 1):
            xx, yy = ix + xi, iy + yi
            if 0 <= xx < gw and 0 <= yy < gh:
                nb[xi + 4 + yi * 3] = xx + yy * gw
    return nb


def _left_cells(n1, grid_type):  # getGridIndexLeft, DLL@0x180047bc0
    fx = (np.float32(LEFT) * n1[:, 0]).astype(np.float32)  # mulss -> fp32
    fy = (np.float32(LEFT) * n1[:, 1]).astype(np.float32)
    x = np.floor(fx.astype(np.float64) + 0.5) if grid_type in (2, 4) else np.floor(fx)
    y = np.floor(fy.astype(np.float64) + 0.5) if grid_type in (3, 4) else np.floor(fy)
    x, y = x.astype(np.int64), y.astype(np.int64)
    cell = x + y * LEFT
    cell[(x >= LEFT) | (y >= LEFT)] = -1  # one common test, no lower bound
    return cell


def _right_cells(n2, wr, hr):  # getGridIndexRight, DLL@0x180047d60
    x = np.floor((np.float32(wr) * n2[:, 0]).astype(np.float32)).astype(np.int64)
    y = np.floor((np.float32(hr) * n2[:, 1]).astype(np.float32)).astype(np.int64)
    return x + y * wr


def match_mask(size1, size2, xy1, xy2, query, train, with_rotation=False, with_scale=False, threshold_factor=6.0):
    """Returns (mask uint8[m], best_scale, best_rot). xy1/xy2: float32 [n, 2] keypoint positions."""
    xy1 = np.asarray(xy1, dtype=np.float32).reshape(-1, 2)
    xy2 = np.asarray(xy2, dtype=np.float32).reshape(-1, 2)
    query = np.asarray(query, dtype=np.int64)
    train = np.asarray(train, dtype=np.int64)
    m = len(query)
    # normalizePoints, DLL@0x180048420: fp32 divide by the int extent converted to fp32
    n1 = np.stack([xy1[:, 0] / np.float32(size1[0]), xy1[:, 1] / np.float32(size1[1])], axis=1).astype(np.float32)
    n2 = np.stack([xy2[:, 0] / np.float32(size2[0]), xy2[:, 1] / np.float32(size2[1])], axis=1).astype(np.float32)
    p1, p2 = n1[query], n2[train]
    nb_left = [_neighbors(i, LEFT, LEFT) for i in range(LEFT * LEFT)]
    left = {g: _left_cells(p1, g) for g in (1, 2, 3, 4)}

    best_mask, best_count, best_scale, best_rot = np.zeros(m, dtype=np.uint8), 0, -1, -1
    for scale in (range(5) if with_scale else (0,)):
        wr = _round_half_even(LEFT * SCALE_RATIOS[scale])
        hr = wr
        nb_right = [_neighbors(j, wr, hr) for j in range(wr * hr)]
        right = _right_cells(p2, wr, hr)
        assert m == 0 or (right.min() >= 0 and right.max() < wr * hr), "outside the parity domain"
        # rotation only enters the neighbour sums: bin once per (scale, grid type)
        binned = {}
        for g in (1, 2, 3, 4):
            lc = left[g]
            keep = lc >= 0
            counts = Counter(zip(lc[keep].tolist(), right[keep].tolist()))
            n_left = Counter(lc[keep].tolist())
            # arg-max per left cell: highest count, lowest right cell on ties
            argmax = {}
            for (l, r), c in counts.items():
                cur = argmax.get(l)
                if cur is None or c > cur[0] or (c == cur[0] and r < cur[1]):
                    argmax[l] = (c, r)
            binned[g] = (lc, counts, n_left, argmax)
        for rot in (range(1, 9) if with_rotation else (1,)):
            pattern = ROTATION_PATTERNS[rot - 1]
            mask = np.zeros(m, dtype=bool)
            for g in (1, 2, 3, 4):
                lc, counts, n_left, argmax = binned[g]
                cell_pair = {}
                for l, (_, jstar) in argmax.items():
                    score, total, pairs = 0, 0.0, 0
                    for k in range(9):
                        ll, rr = nb_left[l][k], nb_right[jstar][pattern[k] - 1]
                        if ll == -1 or rr == -1:
                            continue
                        score += counts.get((ll, rr), 0)
                        total += float(n_left.get(ll, 0))
                        pairs += 1
                    thresh = math.sqrt(total / float(pairs)) * threshold_factor  # divsd, sqrtsd, mulsd
                    cell_pair[l] = -2 if thresh > float(score) else jstar
                want = np.array([cell_pair.get(int(l), -1) if l >= 0 else -3 for l in lc], dtype=np.int64)
                mask |= (lc >= 0) & (want == right)
            count = int(mask.sum())
            if count > best_count:  # strict '>': first hypothesis wins ties
                best_mask, best_count, best_scale, best_rot = mask.astype(np.uint8), count, scale, rot
    return best_mask, best_scale, best_rot
