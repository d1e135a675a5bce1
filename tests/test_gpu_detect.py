"""-m gpu: the keypoint source (gms_detect_batch_device / gms_describe_device) against its definition oracle/detect_ref.c -- records,
order and descriptor bits exact -- and the reference's disparity demo (DisparityUtil.cpp:93-201 with alg "GMS") run from real pixels:
detect -> BFMatcher::match -> matchGMS -> disparity map + RMS against the ground truth, every stage on the GPU, every stage compared."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "image_stereo_pair_450x375.npz")


def _same(got_kp, got_rows, want_kp, want_rows):
    assert len(got_kp) == len(want_kp)
    assert got_kp.tobytes() == want_kp.tobytes()
    assert got_rows.tobytes() == want_rows.tobytes()


@pytest.mark.parametrize("threshold,max_kp", [(20, 10000), (8, 700), (40, 10000), (20, 1), (5, 4000), (254, 50)])
def test_detect_real_pair_matches_the_definition(ctx, pkg, oracle, threshold, max_kp):
    batch = importlib.import_module("sfm-gms_amd.batch")
    z = np.load(GOLDEN)
    imgs = np.stack([z["left"], z["right"]])
    kps, rows = batch.detect_images(ctx, imgs, threshold, max_kp)
    for i in range(2):
        want_kp, want_rows = oracle.detect(imgs[i], threshold, max_kp)
        _same(kps[i], rows[i], want_kp, want_rows)
    assert max_kp == 1 or threshold == 254 or len(kps[0]) > 100


@pytest.mark.parametrize("w,h", [(33, 33), (97, 65), (64, 48), (130, 200), (641, 479), (1030, 50)])
def test_detect_odd_sizes_noise_and_ties(ctx, pkg, oracle, w, h):
    batch = importlib.import_module("sfm-gms_amd.batch")
    rng = np.random.default_rng(w * 1000 + h)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    blocks = (np.kron(rng.integers(0, 2, ((h + 3) // 4, (w + 3) // 4)), np.ones((4, 4), dtype=np.int64))[:h, :w] * 90 + 60).astype(np.uint8)   # two grey levels: every score ties
    sparse = np.full((h, w), 40, dtype=np.uint8)
    sparse[rng.integers(0, h, 60), rng.integers(0, w, 60)] = 200
    imgs = np.stack([noise, blocks, sparse])
    for threshold, max_kp in ((10, 10000), (30, 37), (0, 5)):
        kps, rows = batch.detect_images(ctx, imgs, threshold, max_kp)
        for i in range(3):
            want_kp, want_rows = oracle.detect(imgs[i], threshold, max_kp)
            _same(kps[i], rows[i], want_kp, want_rows)


def test_detect_argument_checks(ctx, pkg):
    import torch
    d = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(pkg.GmsError):
        ctx.detect_batch_device(d.data_ptr(), 1, 32, 100, 20, 10, d.data_ptr(), 1 << 20, d.data_ptr(), d.data_ptr(), d.data_ptr())     # no room for a keypoint
    with pytest.raises(pkg.GmsError):
        ctx.detect_batch_device(d.data_ptr(), 1, 100, 100, 255, 10, d.data_ptr(), 1 << 20, d.data_ptr(), d.data_ptr(), d.data_ptr())   # threshold
    with pytest.raises(pkg.GmsError):
        ctx.detect_batch_device(d.data_ptr(), 1, 100, 100, 20, 10, d.data_ptr(), 1000, d.data_ptr(), d.data_ptr(), d.data_ptr())        # workspace
    ctx.detect_batch_device(d.data_ptr(), 0, 100, 100, 20, 10, None, 0, None, None, None)                                                # nothing to do


def test_describe_every_interior_pixel(ctx, pkg, oracle):
    """Feature2D::compute with a keypoint per pixel (DisparityUtil.cpp:123-133), column by column as the reference builds them."""
    batch = importlib.import_module("sfm-gms_amd.batch")
    img = np.load(GOLDEN)["left"]
    h, w = img.shape
    b = pkg.GMS_DETECT_BORDER
    xs, ys = np.meshgrid(np.arange(b, w - b), np.arange(b, h - b), indexing="ij")     # i (columns) outer, j (rows) inner
    kp = np.zeros(xs.size, dtype=pkg.KEYPOINT_DTYPE)
    kp["x"], kp["y"], kp["size"] = xs.ravel(), ys.ravel(), 1.0
    status, got_kp, got_rows = batch.describe_image(ctx, img, kp)
    rc, want_kp, want_rows = oracle.describe(img, kp)
    assert status == 0 and rc == len(kp)
    _same(got_kp, got_rows, want_kp, want_rows)
    bad = kp[:10].copy()
    bad["y"][4] = b - 1
    status, got_kp, got_rows = batch.describe_image(ctx, img, bad)
    assert status == 1 and oracle.describe(img, bad)[0] == -1


def test_disparity_demo_from_real_pixels(ctx, pkg, oracle, synth):
    """main.cpp's disparity demo with alg "GMS", sparse: both images -> keypoints + rows -> one match per left keypoint -> matchGMS
    (default flags, DisparityUtil.cpp:149) -> disparity map and RMS against left_gt1 / 4 (DisparityUtil.cpp:179-201, 434)."""
    import torch
    batch = importlib.import_module("sfm-gms_amd.batch")
    z = np.load(GOLDEN)
    imgs = np.stack([z["left"], z["right"]])
    h, w = z["left"].shape
    kps, rows = batch.detect_images(ctx, imgs, 12, 10000)
    table = batch.FrameTable(ctx, kps, [(w, h)] * 2)
    dt = batch.DescriptorTable(ctx, table, rows, pkg.GMS_DESC_HAMMING256)
    pairs = np.zeros(1, dtype=pkg.PAIR_DTYPE)
    pairs[0] = (0, 1, len(kps[0]), 0, 0)
    matches = batch.match_pairs(ctx, dt, pairs)
    assert matches.tobytes() == oracle.bf_match(rows[0], rows[1], True).tobytes()
    out, res, mask = batch.filter_pairs(ctx, table, pairs, matches)
    rc, want, want_mask, want_res = oracle.match((w, h), (w, h), kps[0], kps[1], matches)
    n = int(res["n_inliers"][0])
    assert rc == 0 and n == len(want) and out[:n].tobytes() == want.tobytes() and n > 150
    # most survivors of a rectified pair lie on (nearly) the same row, shifted left
    q, t = kps[0][out["queryIdx"][:n]], kps[1][out["trainIdx"][:n]]
    assert (np.abs(q["y"] - t["y"]) <= 2).mean() > 0.9 and ((q["x"] - t["x"]) >= 0).mean() > 0.9
    types = importlib.import_module("sfm-gms_amd.types")
    dev = table.device
    d_matches = batch._to_dev(out[:n], dev)
    d_n = torch.tensor([n], dtype=torch.int32, device=dev)
    d_gt = torch.from_numpy(z["gt"]).to(dev)
    d_disp = torch.zeros(w * h, dtype=torch.uint8, device=dev)
    d_work = torch.zeros(w * h, dtype=torch.int32, device=dev)
    d_stats = torch.zeros(24, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.disparity_device(table.d_kp.data_ptr(), len(kps[0]), table.d_kp.data_ptr() + len(kps[0]) * 28, len(kps[1]), d_matches.data_ptr(),
                         d_n.data_ptr(), n, w, h, d_gt.data_ptr(), 4, d_disp.data_ptr(), d_work.data_ptr(), d_stats.data_ptr())
    ctx.synchronize()
    rc, want_disp, cnt, ssq, mx, rms = oracle.disparity(kps[0], kps[1], want, w, h, z["gt"], 4)
    stats = d_stats.cpu().numpy().view(types.DISPARITY_STATS_DTYPE)[0]
    assert rc == 0 and np.array_equal(d_disp.cpu().numpy().reshape(h, w), want_disp)
    assert (int(stats["count"]), int(stats["sum_sq"]), int(stats["max_abs"]), int(stats["status"])) == (cnt, ssq, mx, 0)
    assert cnt == n or cnt > 0
    assert rms < 6.0      # sparse matches of a clean stereo pair: a few disparity levels off at most


@pytest.mark.parametrize("dense", [False, True])
def test_image_pair_tool(dense):
    """tools/gms_image_pair.py in a fresh process: pixels -> keypoints -> matches -> matchGMS -> disparity RMS, each stage checked against
    the CPU statement (--dense: a keypoint per interior pixel, 143 374 x 143 374 Hamming matches, M = 143 374 into the filter)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "gms_image_pair.py"), "--check"] + (["--dense"] if dense else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(line["check_vs_oracle"].values()), line
    assert line["status"] == 0 and line["survivors"] > (20000 if dense else 1000)
    assert line["keypoints"][0] == (418 * 343 if dense else line["keypoints"][0])
    assert line["disparity_rms"] is not None and line["disparity_rms"] < 12.0
