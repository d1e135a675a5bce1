"""A dataset file in, filtered matches (and two-view geometry) out: the reference's SIFT_matchGMS + structureFromMotion flow
(FeatureMatchUtil.cpp:52-84 -> SfMUtil.cpp:17-82) for every pair of a GMSFRM01 file, on the GPU end to end:

    descriptors --gms_bfmatch_device--> putative matches --gms_filter_device--> survivors
        --gms_two_view_batch_device--> essential matrix, pose, triangulated points, reprojection error   (with a camera)

torch here is device memory only; every stage is a call into csrc/libgms_hip.so. Used by tools/gms_filter_file.py and the tests."""
import numpy as np
import torch

from .batch import DescriptorTable, FrameTable, _to_dev
from .types import DMATCH_DTYPE, PAIR_DTYPE, RESULT_DTYPE, TWO_VIEW_DTYPE, make_camera


def run_dataset(ctx, ds, withRotation=False, withScale=False, thresholdFactor=6.0, match=None, camera=None, dist=None, prob=0.7,
                ransac_threshold=1.0, max_iters=1000, device="cuda:0"):
    """(prob, ransac_threshold: findEssentialMat's confidence and threshold as the flow this function restates passes them -- SfMUtil.cpp:39:
    RANSAC, 0.7, 1.0 -- not OpenCV's own default of 0.999, which gms_find_essential_batch_device's Python mirror keeps.)
    ds: io.Dataset. match=None: brute-force match when the file carries descriptors and no matches. camera = (fx, fy, cx, cy)
    switches the two-view stage on. Returns a dict of host arrays: pairs, matches (the putative ones), out, results, and with a camera
    two_view (TWO_VIEW_DTYPE per pair), coords1, coords2, mask, points3d -- all per-match arrays laid out by match_off."""
    frames = FrameTable(ctx, ds.frames, ds.sizes, device=device)
    dev = frames.device
    pairs = np.ascontiguousarray(ds.pairs, dtype=PAIR_DTYPE).copy()
    n_pairs = len(pairs)
    counts = np.diff(frames.frame_off_host)
    do_match = (ds.descriptors is not None and len(ds.matches) == 0) if match is None else bool(match)
    if do_match:
        if ds.descriptors is None:
            raise ValueError("the dataset carries no descriptors to match")
        # BFMatcher::match without cross-check: one match per keypoint of the query frame (FeatureMatchUtil.cpp:66-68)
        pairs["m"] = counts[pairs["frame_a"]] if n_pairs else 0
        pairs["match_off"] = np.concatenate([[0], np.cumsum(pairs["m"][:-1])]) if n_pairs else 0
    total_m = int((pairs["match_off"] + pairs["m"]).max()) if n_pairs else 0
    max_m = int(pairs["m"].max()) if n_pairs else 0
    d_pairs = _to_dev(pairs, dev) if n_pairs else torch.zeros(24, dtype=torch.uint8, device=dev)
    if do_match:
        descs = DescriptorTable(ctx, frames, ds.descriptors, ds.desc_kind)
        d_matches = torch.zeros(max(total_m, 1) * 16, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        descs.match_device(d_pairs.data_ptr(), n_pairs, max_m, d_matches.data_ptr())
    else:
        m_host = np.ascontiguousarray(ds.matches, dtype=DMATCH_DTYPE)
        if total_m > len(m_host):
            raise ValueError("a pair's match range lies outside the dataset's match array")
        d_matches = _to_dev(m_host, dev) if len(m_host) else torch.zeros(16, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(max(total_m, 1) * 16, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(max(n_pairs, 1) * 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    ctx.filter_device(frames.d_pts.data_ptr(), frames.d_frame_off.data_ptr(), frames.n_frames, d_pairs.data_ptr(), n_pairs, max_m,
                      d_matches.data_ptr(), d_out.data_ptr(), d_res.data_ptr(), None, withRotation, withScale, thresholdFactor)
    out = dict(pairs=pairs)
    if camera is not None:
        cam = make_camera(camera, dist)
        d_c1 = torch.zeros(max(total_m, 1) * 2, dtype=torch.float32, device=dev)
        d_c2 = torch.zeros(max(total_m, 1) * 2, dtype=torch.float32, device=dev)
        d_mask = torch.zeros(max(total_m, 1), dtype=torch.uint8, device=dev)
        d_p3 = torch.zeros(max(total_m, 1) * 3, dtype=torch.float64, device=dev)
        d_tv = torch.zeros(max(n_pairs, 1) * TWO_VIEW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        ctx.two_view_batch_device(cam, frames.d_kp.data_ptr(), frames.d_frame_off.data_ptr(), frames.n_frames, d_pairs.data_ptr(), n_pairs,
                                  max_m, d_out.data_ptr(), d_res.data_ptr(), d_c1.data_ptr(), d_c2.data_ptr(), d_mask.data_ptr(),
                                  d_p3.data_ptr(), d_tv.data_ptr(), prob, ransac_threshold, max_iters)
        ctx.synchronize()
        out.update(two_view=d_tv.cpu().numpy().view(TWO_VIEW_DTYPE)[:n_pairs], coords1=d_c1.cpu().numpy().reshape(-1, 2)[:total_m],
                   coords2=d_c2.cpu().numpy().reshape(-1, 2)[:total_m], mask=d_mask.cpu().numpy()[:total_m],
                   points3d=d_p3.cpu().numpy().reshape(-1, 3)[:total_m])
    ctx.synchronize()
    out.update(matches=d_matches.cpu().numpy().view(DMATCH_DTYPE)[:total_m], out=d_out.cpu().numpy().view(DMATCH_DTYPE)[:total_m],
               results=d_res.cpu().numpy().view(RESULT_DTYPE)[:n_pairs])
    return out
