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


@pytest.mark.gpu
def test_two_workgroups_per_cu_variant_reproduces_golden():
    """GMS_OCC2=1 selects filter_kernel_occ2 (off by default: measured slower); it must stay bit-exact."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, importlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import golden_util\n"
        "pkg = importlib.import_module('sfm-gms_amd'); ctx = pkg.GmsContext(0)\n"
        "for name in golden_util.names():\n"
        "    c, want = golden_util.load(name)\n"
        "    for (rot, scale), (mask, best) in want.items():\n"
        "        out, res = ctx.match(c['size1'], c['size2'], c['kp1'], c['kp2'], c['matches'], rot, scale, 6.0, return_result=True)\n"
        "        assert out.tobytes() == c['matches'][mask].tobytes(), name\n"
        "        assert (int(res['n_inliers']), int(res['best_scale']), int(res['best_rot'])) == best, name\n"
        "print('occ2 ok')\n" % (root, os.path.join(root, "tests")))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, GMS_OCC2="1"))
    assert res.returncode == 0 and "occ2 ok" in res.stdout, res.stderr[-2000:]
