"""CPU: the brute-force matcher's oracle (oracle/bf_ref.c) against known answers and an independent numpy restatement."""
import numpy as np


def _np_hamming(q, t):
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    return d.argmin(axis=1), d.min(axis=1)      # argmin returns the first minimum


def test_hamming_known_answers_and_first_minimum(oracle):
    q = np.zeros((3, 32), dtype=np.uint8)
    t = np.zeros((5, 32), dtype=np.uint8)
    q[1, 0] = 0xFF
    q[2, :] = 0xFF
    t[0, 31] = 0x01          # distance 1 to q0
    t[1, 0] = 0xFF           # == q1
    t[2, 0] = 0xFF           # duplicate of t1: the lower index must win
    t[3, :] = 0xFF           # == q2
    t[4, :] = 0xFF
    m = oracle.bf_match(q, t, True)
    assert m["queryIdx"].tolist() == [0, 1, 2] and m["imgIdx"].tolist() == [0, 0, 0]
    assert m["trainIdx"].tolist() == [0, 1, 3] and m["distance"].tolist() == [1.0, 0.0, 0.0]
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (333, 32), dtype=np.uint8)
    t[50:60] = t[10:20]      # ties
    q[:10] = t[10:20]
    m = oracle.bf_match(q, t, True)
    idx, dist = _np_hamming(q, t)
    assert m["trainIdx"].tolist() == idx.tolist() and m["distance"].tolist() == dist.astype(np.float32).tolist()
    assert m["trainIdx"][:10].tolist() == list(range(10, 20))


def test_l2_known_answers_first_minimum_and_fp32_order(oracle):
    q = np.zeros((2, 128), dtype=np.float32)
    t = np.zeros((4, 128), dtype=np.float32)
    t[0, 0] = 3.0
    t[0, 1] = 4.0            # distance 5 to q0
    t[1, 5] = 2.0            # distance 2
    t[2, 7] = 2.0            # distance 2 as well: index 1 wins
    q[1, :] = 255.0
    t[3, :] = 255.0
    m = oracle.bf_match(q, t, False)
    assert m["trainIdx"].tolist() == [1, 3] and m["distance"].tolist() == [2.0, 0.0]
    # integer-valued rows (SIFT): every partial sum is exact, so the result equals the integer computation
    rng = np.random.default_rng(2)
    q = rng.integers(0, 256, (150, 128)).astype(np.float32)
    t = rng.integers(0, 256, (260, 128)).astype(np.float32)
    t[200:220] = t[20:40]
    m = oracle.bf_match(q, t, False)
    d2 = ((q[:, None, :].astype(np.int64) - t[None, :, :].astype(np.int64)) ** 2).sum(axis=2)
    assert m["trainIdx"].tolist() == d2.argmin(axis=1).tolist()
    assert m["distance"].tobytes() == np.sqrt(d2.min(axis=1).astype(np.float32)).tobytes()
    # general floats: fp32 sums in index order
    q = rng.normal(0, 1, (40, 128)).astype(np.float32)
    t = rng.normal(0, 1, (70, 128)).astype(np.float32)
    m = oracle.bf_match(q, t, False)
    want = np.zeros((40, 70), dtype=np.float32)
    for k in range(128):
        dd = q[:, None, k] - t[None, :, k]
        want = want + dd * dd
    assert m["trainIdx"].tolist() == want.argmin(axis=1).tolist()
    assert m["distance"].tobytes() == np.sqrt(want.min(axis=1)).tobytes()


def test_no_train_rows(oracle):
    m = oracle.bf_match(np.zeros((3, 32), dtype=np.uint8), np.zeros((0, 32), dtype=np.uint8), True)
    assert m["trainIdx"].tolist() == [-1, -1, -1]
