"""CPU: the C oracle (dense, reference loop order) against the independently written sparse Python
restatement, on random and adversarial inputs and every flag combination."""
import numpy as np
import pytest

import cases
import gms_ref_sparse

ADV = cases.adversarial_cases()


def _cross(oracle, c, rot, scale, thr=6.0):
    rc, out, mask, res = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, thr)
    assert rc == 0
    xy1 = np.stack([c["kp1"]["x"], c["kp1"]["y"]], axis=1)
    xy2 = np.stack([c["kp2"]["x"], c["kp2"]["y"]], axis=1)
    m2, s2, r2 = gms_ref_sparse.match_mask(c["size1"], c["size2"], xy1, xy2, c["matches"]["queryIdx"],
                                           c["matches"]["trainIdx"], rot, scale, thr)
    assert np.array_equal(mask, m2)
    assert (res["best_scale"], res["best_rot"]) == (s2, r2)
    assert out.tobytes() == c["matches"][mask.astype(bool)].tobytes()  # verbatim, input order
    assert res["n_inliers"] == int(mask.sum())
    return int(mask.sum())


@pytest.mark.parametrize("name", sorted(ADV))
@pytest.mark.parametrize("rot,scale", cases.FLAGS)
def test_adversarial(oracle, name, rot, scale):
    _cross(oracle, ADV[name], rot, scale)


@pytest.mark.parametrize("rot,scale", cases.FLAGS)
def test_random_small(oracle, rot, scale):
    kept = _cross(oracle, cases.random_pair(21, n=1500, inlier_frac=0.6), rot, scale)
    assert kept > 100


def test_expected_structure_of_adversarial_cases(oracle):
    """The hand-built cases do what they were built for."""
    def run(name, rot=False, scale=False):
        c = ADV[name]
        rc, out, mask, res = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0)
        assert rc == 0
        return c, mask, res
    # threshold tie is kept, one agreeing match fewer is not (centre cell 210 holds matches 0..3)
    _, mask, _ = run("thresh_tie")
    assert mask[:4].all()
    _, mask, _ = run("thresh_just_below")
    assert not mask[:4].any()
    # arg-max tie: the lower right cell wins, so exactly the "lc - 40" half of each cell survives
    c, mask, _ = run("argmax_tie")
    assert mask.sum() == len(mask) // 2
    assert mask.reshape(-1, 24)[:, 12:].all() and not mask.reshape(-1, 24)[:, :12].any()
    # pure outliers: nothing survives, no hypothesis is selected
    for flags in cases.FLAGS:
        _, mask, res = run("all_outliers", *flags)
        assert mask.sum() == 0 and res["best_scale"] == -1 and res["best_rot"] == -1
    # empty and single inputs
    _, mask, res = run("m0")
    assert len(mask) == 0 and res["n_inliers"] == 0
    _, mask, _ = run("m1")
    assert mask.sum() == 0  # 6 * sqrt(1/9) = 2 > 1


@pytest.mark.parametrize("name", sorted(cases.domain_error_cases()))
def test_domain_errors(oracle, name):
    c = cases.domain_error_cases()[name]
    rc, out, mask, res = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], True, True, 6.0)
    assert rc == -2 and len(out) == 0 and res["status"] == -2


def test_threshold_factor_and_unconditional_mask(oracle):
    c = cases.random_pair(22, n=1200, inlier_frac=0.5)
    kept = []
    for thr in (0.0, 3.0, 6.0, 12.0, 1e9):
        rc, out, mask, _ = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], False, False, thr)
        assert rc == 0
        kept.append(int(mask.sum()))
    assert kept == sorted(kept, reverse=True) and kept[-1] == 0 and kept[0] > kept[2] > 0


def test_oracle_batch_threads_agree(oracle, synth, pkg):
    frames = synth.make_sequence(5, 6, size=(640, 480), n_kp=400)
    kp_all = np.concatenate(frames)
    foff = np.arange(7, dtype=np.int64) * 400
    wh = np.array([640, 480] * 6, dtype=np.int32)
    pairs = np.zeros(10, dtype=pkg.PAIR_DTYPE)
    matches = []
    for i in range(10):
        a, b = pkg.pair_from_index(i, 6)
        mt = synth.sequence_matches(100 + i, 400, 400, 0.6)
        pairs[i] = (a, b, len(mt), 0, i * 400)
        matches.append(mt)
    matches = np.concatenate(matches)
    f1, o1, r1, m1 = oracle.batch(kp_all, foff, wh, pairs, matches, True, False, 6.0, 1)
    f4, o4, r4, m4 = oracle.batch(kp_all, foff, wh, pairs, matches, True, False, 6.0, 4)
    assert f1 == f4 == 0 and np.array_equal(m1, m4) and r1.tobytes() == r4.tobytes() and o1.tobytes() == o4.tobytes()
    assert r1["n_inliers"].sum() > 0


def test_oracle_batch_on_reused_storage_equals_fresh_calls(oracle, synth, pkg):
    """gms_ref_batch keeps one scratch state per thread (gms_ref_match_ws): a thread's pairs alternate between large and small
    shapes and every flag combination, and each must equal the one-shot call on fresh storage."""
    sizes = [(640, 480), (1920, 1080), (333, 777)]
    counts = [700, 60, 1500, 5, 0, 900]
    rng = np.random.default_rng(5)
    frames = [synth.make_keypoints(np.stack([rng.uniform(0, sizes[i % 3][0] - 1, n), rng.uniform(0, sizes[i % 3][1] - 1, n)],
                                            axis=1).astype(np.float32)) for i, n in enumerate(counts)]
    foff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    wh = np.array([sizes[i % 3] for i in range(len(counts))], dtype=np.int32).reshape(-1)
    combos = [(0, 2), (1, 3), (2, 0), (3, 5), (5, 2), (2, 5), (1, 0), (4, 2), (0, 5)]
    pairs = np.zeros(len(combos), dtype=pkg.PAIR_DTYPE)
    chunks, off = [], 0
    for i, (a, b) in enumerate(combos):
        m = 0 if min(counts[a], counts[b]) == 0 else int(rng.integers(1, 2 * counts[a]))
        q = rng.integers(0, max(counts[a], 1), m)
        t = np.where(rng.uniform(size=m) < 0.6, q % max(counts[b], 1), rng.integers(0, max(counts[b], 1), m))
        chunks.append(synth.make_matches(q, t, rng))
        pairs[i] = (a, b, m, 0, off)
        off += m
    matches = np.concatenate(chunks)
    kp_all = np.concatenate(frames)
    for rot, scale in cases.FLAGS:
        for threads in (1, 3):
            failed, out, res, mask = oracle.batch(kp_all, foff, wh, pairs, matches, rot, scale, 6.0, threads)
            assert failed == 0
            for i, (a, b) in enumerate(combos):
                o, m = int(pairs[i]["match_off"]), int(pairs[i]["m"])
                rc, want, wmask, wres = oracle.match(sizes[a % 3], sizes[b % 3], frames[a], frames[b], matches[o:o + m], rot, scale, 6.0)
                assert rc == 0 and res[i].tobytes() == wres.tobytes()
                assert np.array_equal(mask[o:o + m], wmask) and out[o:o + len(want)].tobytes() == want.tobytes()
