"""CPU: host-side logic -- the C-ABI library loads and exports what include/gms.h declares, the product
path fails loudly without a GPU, PODs have the reference's layout, sharding tiles the pair list, the
synthetic recipe is deterministic, and the N>1 driver logic works under gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "gms.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gms_[a-z_0-9]+)\s*\(", hdr)))
    assert sorted(pkg.EXPORTED_SYMBOLS) == declared
    lib = pkg.load_library()  # no compute call: loading and symbol lookup work without a GPU
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.gms_version()
    assert lib.gms_max_matches() >= 10000
    assert lib.gms_error_string(-2).decode().startswith("input outside")


def test_pod_layouts_match_reference_strides(pkg):
    assert pkg.KEYPOINT_DTYPE.itemsize == 0x1C and pkg.DMATCH_DTYPE.itemsize == 0x10
    assert pkg.KEYPOINT_DTYPE.fields["y"][1] == 4 and pkg.DMATCH_DTYPE.fields["trainIdx"][1] == 4
    assert pkg.PAIR_DTYPE.itemsize == 24 and pkg.PAIR_DTYPE.fields["match_off"][1] == 16


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this box has a GPU")
def test_product_path_fails_loudly_without_gpu(pkg, synth):
    """No CPU fallback: without a device the compute entry points return an error / raise."""
    lib = pkg.load_library()
    h = C.c_void_p()
    assert lib.gms_ctx_create(0, C.byref(h)) == -4  # GMS_ERR_NO_DEVICE
    kp1, kp2, m = synth.make_pair(1, size1=(640, 480), n1=50)
    with pytest.raises(pkg.GmsError):
        pkg.matchGMS((640, 480), (640, 480), kp1, kp2, m)
    out = np.zeros(50, dtype=pkg.DMATCH_DTYPE)
    n_out = C.c_int(123)
    rc = lib.gms_match(kp1.ctypes.data, 50, 640, 480, kp2.ctypes.data, 50, 640, 480, m.ctypes.data, 50, 0, 0, 6.0,
                       out.ctypes.data, C.byref(n_out))
    assert rc == -4 and n_out.value == 0


def test_missing_library_is_an_import_error(pkg, monkeypatch):
    capi = sys.modules["sfm-gms_amd.capi"]
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "library_path", lambda: "/nonexistent/libgms_hip.so")
    with pytest.raises(ImportError):
        capi.load_library()


def test_product_sources_never_touch_the_oracle():
    pkg_dir = os.path.join(ROOT, "sfm-gms_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "gms_ref" not in text and "gms_oracle" not in text and "oracle/" not in text, f


def test_pair_indexing_and_shards(pkg):
    for n in (2, 3, 7, 50):
        total = pkg.all_pairs_count(n)
        seen = [pkg.pair_from_index(k, n) for k in range(total)]
        assert seen == [(a, b) for a in range(n) for b in range(a + 1, n)]
    assert pkg.all_pairs_count(1000) == 499500 and pkg.pair_from_index(499499, 1000) == (998, 999)
    with pytest.raises(IndexError):
        pkg.pair_from_index(3, 3)
    for n_items in (0, 1, 7, 4096, 499500):
        for world in (1, 2, 3, 8):
            cuts = [pkg.shard_range(n_items, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n_items
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_recipe_is_deterministic(synth):
    a = synth.make_pair(3, size1=(640, 480), n1=200)
    b = synth.make_pair(3, size1=(640, 480), n1=200)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
    kp1, kp2, m = a
    assert (m["queryIdx"] == np.arange(200)).all()  # BFMatcher without cross-check: M = N1, queryIdx = i
    assert kp1["x"].min() >= 0 and kp1["x"].max() < 639 and kp2["y"].max() < 479
    f1 = synth.make_sequence(9, 4, size=(640, 480), n_kp=100)
    f2 = synth.make_sequence(9, 4, size=(640, 480), n_kp=100)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(f1, f2))


_GLOO_WORKER = r'''
# The rank-side driver logic of bench.py (sfm-gms_amd/dist.py: RankPlan, synth_matches_device, parity_sample and the gloo
# rendezvous), on CPU tensors: everything except the filter launch itself.
import importlib, os, sys, json
sys.path.insert(0, sys.argv[1])
import numpy as np
pkg = importlib.import_module("sfm-gms_amd")
d = importlib.import_module("sfm-gms_amd.dist")
rank, local_rank, world = d.env_world()
dist = d.init_rendezvous(world)
assert dist is not None and dist.get_backend() == "gloo"
n_frames, n_kp, per_step, n_chunks = 40, 300, 64, 5
plan = d.RankPlan(n_frames, n_kp, per_step, n_chunks, rank, world)
assert plan.total_pairs == 780 and (plan.lo, plan.hi) == pkg.shard_range(780, rank, world)
seen = []
for c in range(n_chunks):
    t = plan.chunk_pairs(c)
    k0 = plan.starts[c]
    assert plan.lo <= k0 and k0 + plan.chunk <= plan.hi          # a chunk never leaves the rank's block
    assert [(int(a), int(b)) for a, b in zip(t["frame_a"], t["frame_b"])] == [pkg.pair_from_index(k0 + j, n_frames) for j in range(plan.chunk)]
    assert (t["match_off"] == np.arange(plan.chunk) * n_kp).all() and (t["m"] == n_kp).all()
    m = d.synth_matches_device(k0, plan.chunk, n_kp, 0.5, "cpu").numpy().view(np.uint8).reshape(-1).view(pkg.DMATCH_DTYPE)
    for k, j in plan.chunk_sample(c):
        assert k % d.PARITY_EVERY == 0 and k == k0 + j
        seen.append(k)
    j = plan.chunk // 2                                           # any pair: device form == host form, whatever the rank
    assert m[j * n_kp:(j + 1) * n_kp].tobytes() == d.synth_matches_host(k0 + j, n_kp, 0.5).tobytes()
d.barrier(dist)
slow = d.max_over_ranks(1.0 + rank, dist)
tot = d.sum_over_ranks([plan.hi - plan.lo, len(set(seen))], dist)
counts = d.gather_counts(plan.hi - plan.lo, dist)
if rank == 0:
    print(json.dumps({"max": slow, "counts": counts, "sum": tot, "starts": plan.starts}))
d.barrier(dist)
dist.destroy_process_group()
'''


def test_two_rank_driver_logic_under_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29731", str(script), ROOT]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    import json
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    got = json.loads(line)
    assert got["max"] == 2.0 and got["counts"] == [390, 390] and got["sum"][0] == 780
    assert got["starts"] == [0, 64, 128, 192, 256]   # rank 0 walks its block front to back
    assert got["sum"][1] == 1                         # pair 0 is the only multiple of 997 below 780 (rank 0 filtered it)


def test_bench_imports_the_driver_module_the_gloo_test_covers():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name in ("RankPlan", "synth_matches_device", "init_rendezvous", "max_over_ranks", "sum_over_ranks", "barrier"):
        assert ("distmod." + name in src) or ("self.dist." + name in src) or ("wl.dist." + name in src), name
    assert '"nccl"' not in src and "init_process_group" not in src  # the rendezvous is dist.py's (gloo)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                         timeout=120, env=env)
    assert res.returncode != 0 and "--gpus 4" in res.stderr


def test_vectorised_pair_indexing_and_match_synthesis(pkg):
    import importlib
    d = importlib.import_module("sfm-gms_amd.dist")
    for n in (2, 3, 17, 1000):
        total = pkg.all_pairs_count(n)
        k = np.unique(np.concatenate([np.arange(min(total, 400)), np.arange(max(0, total - 400), total),
                                      np.linspace(0, total - 1, 300).astype(np.int64)]))
        a, b = d.pairs_from_indices(k, n)
        assert [pkg.pair_from_index(int(x), n) for x in k] == list(zip(a.tolist(), b.tolist()))
    with pytest.raises(IndexError):
        d.pairs_from_indices([3], 3)
    m = d.synth_matches_host(499499, 10000, 0.5)
    assert (m["queryIdx"] == np.arange(10000)).all() and m["trainIdx"].min() >= 0 and m["trainIdx"].max() < 10000
    assert 0.47 < (m["trainIdx"] == m["queryIdx"]).mean() < 0.53 and 0 <= m["distance"].min() and m["distance"].max() < 256
    dev = d.synth_matches_device(499498, 2, 10000, 0.5, "cpu").numpy().view(np.uint8).reshape(-1).view(pkg.DMATCH_DTYPE)
    assert dev[10000:].tobytes() == m.tobytes()
    assert d.parity_sample(0, 2000) == [0, 997, 1994] and d.parity_sample(998, 996) == [] and d.parity_sample(998, 997) == [1994]
    starts, chunk = d.chunk_starts(100, 1100, 300, 5)
    assert chunk == 300 and starts == [100, 400, 700, 100, 400]
    starts, chunk = d.chunk_starts(0, 50, 300, 2)
    assert chunk == 50 and starts == [0, 0]
