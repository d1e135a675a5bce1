"""Device-resident batch driver: resident frame tables + pair descriptors -> filtered matches.

torch is used here for what the project allows it for: device memory and streams. The filter itself
is gms_filter_device in csrc/libgms_hip.so. Mirrors how the reference's callers use matchGMS
(FeatureMatchUtil.cpp:66-69: M = N1 matches from BFMatcher, then one matchGMS per image pair), but
for many pairs per launch with the per-frame keypoint tables kept in HBM.
"""
import numpy as np
import torch

from .api import GmsContext
from .types import DMATCH_DTYPE, KEYPOINT_DTYPE, PAIR_DTYPE, RESULT_DTYPE


def _to_dev(arr, device):
    a = np.ascontiguousarray(arr)
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(device)


class FrameTable:
    """Keypoints of all frames of a sequence, normalised once on the GPU (GMSMatcher::normalizePoints)."""

    def __init__(self, ctx, keypoints_per_frame, sizes, device="cuda:0"):
        self.ctx = ctx
        self.device = torch.device(device)
        self.n_frames = len(keypoints_per_frame)
        counts = np.array([len(k) for k in keypoints_per_frame], dtype=np.int64)
        self.frame_off_host = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        self.total = int(self.frame_off_host[-1])
        kp_all = (np.concatenate([np.ascontiguousarray(k, dtype=KEYPOINT_DTYPE) for k in keypoints_per_frame])
                  if self.total else np.zeros(0, dtype=KEYPOINT_DTYPE))
        wh = np.asarray(sizes, dtype=np.int32).reshape(-1, 2)
        assert wh.shape[0] == self.n_frames
        self.d_kp = _to_dev(kp_all, self.device)
        self.d_frame_off = torch.from_numpy(self.frame_off_host).to(self.device)
        self.d_wh = torch.from_numpy(wh.reshape(-1).copy()).to(self.device)
        # the frame table: normalised points, then the per-keypoint cell codes (gms_frame_table_bytes)
        self.d_pts = torch.zeros(ctx.frame_table_bytes(self.total) // 4, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        ctx.normalize_device(self.d_kp.data_ptr(), self.d_frame_off.data_ptr(), self.d_wh.data_ptr(),
                             self.n_frames, self.total, self.d_pts.data_ptr())
        ctx.synchronize()


class DescriptorTable:
    """Descriptors of all frames of a sequence, resident on the GPU beside a FrameTable (descriptor i of a frame belongs to
    keypoint i). kind = GMS_DESC_HAMMING256: uint8 [n, 32] rows (ORB); GMS_DESC_L2_F32X128: float32 [n, 128] rows (SIFT), for
    which the per-frame tables of the matcher are prepared once here (gms_bf_prepare_device)."""

    def __init__(self, ctx, frames, descriptors_per_frame, kind):
        self.ctx, self.kind, self.frames = ctx, int(kind), frames
        dt, width = (np.uint8, 32) if self.kind == 0 else (np.float32, 128)
        rows = [np.ascontiguousarray(d, dtype=dt).reshape(-1, width) for d in descriptors_per_frame]
        assert [len(r) for r in rows] == list(np.diff(frames.frame_off_host)), "one descriptor per keypoint"
        self.host = np.concatenate(rows) if frames.total else np.zeros((0, width), dtype=dt)
        self.d_desc = torch.from_numpy(self.host.view(np.uint8).reshape(-1)).to(frames.device) if frames.total else \
            torch.zeros(16, dtype=torch.uint8, device=frames.device)
        nbytes = ctx.bf_prepared_bytes(self.kind, frames.total, frames.n_frames)
        self.d_prep = torch.zeros(max(nbytes, 16), dtype=torch.uint8, device=frames.device)
        torch.cuda.synchronize(frames.device)
        ctx.bf_prepare_device(self.kind, self.d_desc.data_ptr(), frames.d_frame_off.data_ptr(), frames.n_frames, frames.total,
                              self.d_prep.data_ptr())
        ctx.synchronize()

    def match_device(self, d_pairs, n_pairs, max_query, d_matches, use_prepared=True):
        """gms_bfmatch_device: one match per query row of every pair, written at the pair's match_off (stream-ordered).
        use_prepared=False (Hamming only): the vector-ALU kernel on the raw rows instead of the matrix-core one."""
        f = self.frames
        self.ctx.bfmatch_device(self.kind, self.d_desc.data_ptr(), self.d_prep.data_ptr() if use_prepared else None, f.total,
                                f.d_frame_off.data_ptr(), f.n_frames, d_pairs, n_pairs, max_query, d_matches)


def detect_images(ctx, images, threshold=20, max_keypoints=10000, device=None):
    """gms_detect_batch_device on a stack of equally sized 8-bit grey images [n, H, W] (host array or device tensor): returns
    (keypoints_per_image, rows_per_image) as host arrays -- KEYPOINT_DTYPE records in raster order and uint8 [n_i, 32] rows."""
    dev = torch.device(device if device is not None else f"cuda:{ctx.device}")
    imgs = images if torch.is_tensor(images) else torch.from_numpy(np.ascontiguousarray(images, dtype=np.uint8))
    if imgs.dim() == 2:
        imgs = imgs[None]
    imgs = imgs.to(dev).contiguous()
    n, h, w = imgs.shape
    nb = ctx.detect_workspace_bytes(w, h, n, max_keypoints)
    if nb == 0:
        raise ValueError("bad image size")
    d_ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    d_kp = torch.zeros(max(n * max_keypoints, 1) * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(max(n * max_keypoints, 1) * 32, dtype=torch.uint8, device=dev)
    d_counts = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    ctx.detect_batch_device(imgs.data_ptr(), n, w, h, threshold, max_keypoints, d_ws.data_ptr(), nb, d_kp.data_ptr(), d_desc.data_ptr(),
                            d_counts.data_ptr())
    ctx.synchronize()
    counts = d_counts.cpu().numpy()
    kp = d_kp.cpu().numpy()[: n * max_keypoints * 28].view(KEYPOINT_DTYPE).reshape(n, max_keypoints)
    desc = d_desc.cpu().numpy()[: n * max_keypoints * 32].reshape(n, max_keypoints, 32)
    return [kp[i, : counts[i]].copy() for i in range(n)], [desc[i, : counts[i]].copy() for i in range(n)]


def describe_image(ctx, image, keypoints, device=None):
    """gms_describe_device: directions and 32-byte rows at the given integer keypoints of one image -> (status, keypoints, rows)."""
    dev = torch.device(device if device is not None else f"cuda:{ctx.device}")
    img = torch.from_numpy(np.ascontiguousarray(image, dtype=np.uint8)).to(dev)
    h, w = img.shape
    kp = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE)
    nb = ctx.detect_workspace_bytes(w, h, 1, 0)
    if nb == 0:
        raise ValueError("bad image size")
    d_ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    d_kp = _to_dev(kp, dev) if len(kp) else torch.zeros(28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(max(len(kp), 1) * 32, dtype=torch.uint8, device=dev)
    d_status = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    ctx.describe_device(img.data_ptr(), w, h, d_kp.data_ptr(), len(kp), d_ws.data_ptr(), nb, d_desc.data_ptr(), d_status.data_ptr())
    ctx.synchronize()
    out_kp = d_kp.cpu().numpy()[: len(kp) * 28].view(KEYPOINT_DTYPE).copy() if len(kp) else kp
    return int(d_status.item()), out_kp, d_desc.cpu().numpy()[: len(kp) * 32].reshape(-1, 32)


def match_pairs(ctx, descs, pairs, use_prepared=True):
    """Brute-force matches of `pairs` (PAIR_DTYPE; m = keypoints of frame_a) as a host DMATCH_DTYPE array laid out by match_off."""
    dev = descs.frames.device
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    total = int((pairs["match_off"] + pairs["m"]).max()) if len(pairs) else 0
    d_pairs = _to_dev(pairs, dev)
    d_matches = torch.zeros(max(total, 1) * 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    descs.match_device(d_pairs.data_ptr(), len(pairs), int(pairs["m"].max()) if len(pairs) else 0, d_matches.data_ptr(), use_prepared)
    ctx.synchronize()
    return d_matches.cpu().numpy().view(DMATCH_DTYPE)[:total]


def filter_pairs(ctx, frames, pairs, matches, withRotation=False, withScale=False, thresholdFactor=6.0,
                 want_mask=True):
    """Run the filter over `pairs` (PAIR_DTYPE array) whose matches live in `matches` (DMATCH_DTYPE array,
    pair i at [match_off, match_off+m)). Returns (out, results, mask) as host arrays."""
    dev = frames.device
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    matches = np.ascontiguousarray(matches, dtype=DMATCH_DTYPE)
    n_pairs = len(pairs)
    max_m = int(pairs["m"].max()) if n_pairs else 0
    total_m = len(matches)
    d_pairs = _to_dev(pairs, dev)
    d_matches = _to_dev(matches, dev) if total_m else torch.zeros(16, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(max(total_m, 1) * 16, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(max(n_pairs, 1) * 16, dtype=torch.uint8, device=dev)
    d_mask = torch.zeros(max(total_m, 1), dtype=torch.uint8, device=dev) if want_mask else None
    torch.cuda.synchronize(dev)
    ctx.filter_device(frames.d_pts.data_ptr(), frames.d_frame_off.data_ptr(), frames.n_frames,
                      d_pairs.data_ptr(), n_pairs, max_m, d_matches.data_ptr(), d_out.data_ptr(),
                      d_res.data_ptr(), d_mask.data_ptr() if want_mask else None,
                      withRotation, withScale, thresholdFactor)
    ctx.synchronize()
    out = d_out.cpu().numpy().view(DMATCH_DTYPE)[:total_m]
    res = d_res.cpu().numpy().view(RESULT_DTYPE)[:n_pairs]
    mask = d_mask.cpu().numpy()[:total_m] if want_mask else None
    return out, res, mask
