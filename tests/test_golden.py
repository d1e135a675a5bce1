"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py).
CPU: the oracle reproduces them. GPU (-m gpu): the HIP path reproduces them through the C ABI."""
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
