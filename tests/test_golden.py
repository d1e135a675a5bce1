"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py).
CPU: the oracle reproduces them. GPU (-m gpu): the HIP path reproduces them through the C ABI.
tests/golden/refdll_chain.npz holds, for the same inputs, what THE REFERENCE BINARY computes: matchGMS's computation run as a chain of
the DLL's own pieces (tests/golden/refdll_runner.c "chain": normalizePoints, getInlierMask whole, setScale's head,
initalizeNeighbors, assignMatchPairs, the body of verifyCellPairs, the marking / counting tail of run; the glue between the pieces --
storage, the zero fills, the row-sum test -- is the runner's). Oracle and HIP path are compared with that as well."""
import os

import numpy as np
import pytest

import cases
import golden_util

NAMES = golden_util.names()


def test_fixtures_exist():
    assert len(NAMES) >= 10


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(oracle, name):
    c, want = golden_util.load(name)
    for (rot, scale), (mask, best) in want.items():
        rc, out, got_mask, res = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0)
        assert rc == 0
        assert (got_mask.astype(bool) == mask).all()
        assert (int(res["n_inliers"]), int(res["best_scale"]), int(res["best_rot"])) == best


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(ctx, name):
    c, want = golden_util.load(name)
    for (rot, scale), (mask, best) in want.items():
        out, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], rot, scale, 6.0, return_result=True)
        assert out.tobytes() == c["matches"][mask].tobytes()
        assert (int(res["n_inliers"]), int(res["best_scale"]), int(res["best_rot"])) == best


def _chain():
    return np.load(os.path.join(golden_util.GOLDEN_DIR, "refdll_chain.npz"))


def test_reference_chain_covers_every_golden_case():
    z = _chain()
    for name in NAMES:
        for tag in ("r0s0", "r0s1", "r1s0", "r1s1"):
            assert f"{name}_{tag}_mask" in z.files and f"{name}_{tag}_ret" in z.files
    assert "config2_1080p_10k" in NAMES and "config1_640x480_500" in NAMES      # BASELINE configs 1 and 2


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_the_reference_binary_chain(oracle, name):
    """end to end: the mask and the count the DLL's own code produces on this case, all four flag combinations"""
    c, _ = golden_util.load(name)
    z = _chain()
    m = len(c["matches"])
    for rot in (0, 1):
        for scale in (0, 1):
            rc, out, got_mask, res = oracle.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], bool(rot), bool(scale), 6.0)
            want = np.unpackbits(z[f"{name}_r{rot}s{scale}_mask"], bitorder="little")[:m]
            assert rc == 0 and np.array_equal(got_mask, want)
            assert int(res["n_inliers"]) == int(z[f"{name}_r{rot}s{scale}_ret"]) == int(want.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_equals_the_reference_binary_chain(ctx, name):
    c, _ = golden_util.load(name)
    z = _chain()
    m = len(c["matches"])
    for rot in (0, 1):
        for scale in (0, 1):
            out, res = ctx.match(c["size1"], c["size2"], c["kp1"], c["kp2"], c["matches"], bool(rot), bool(scale), 6.0, return_result=True)
            want = np.unpackbits(z[f"{name}_r{rot}s{scale}_mask"], bitorder="little")[:m].astype(bool)
            assert out.tobytes() == c["matches"][want].tobytes()
            assert int(res["n_inliers"]) == int(z[f"{name}_r{rot}s{scale}_ret"])
