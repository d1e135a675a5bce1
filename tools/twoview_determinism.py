#!/usr/bin/env python3
"""Diagnostic: gms_two_view_batch_device run several times on the same 1024 filtered pairs -- are the records bit-identical from run
to run, and do they equal the host build of the same arithmetic (tests/cpp/twoview_host.cpp) pair by pair?"""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    pkg = importlib.import_module("sfm-gms_amd")
    synth = importlib.import_module("sfm-gms_amd.synth")
    batch = importlib.import_module("sfm-gms_amd.batch")
    types = importlib.import_module("sfm-gms_amd.types")
    distmod = importlib.import_module("sfm-gms_amd.dist")
    so = "/tmp/libtvh.so"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "sfm-gms_amd", "csrc"),
                           "-o", so, os.path.join(ROOT, "tests", "cpp", "twoview_host.cpp")])
    tvh = C.CDLL(so)
    vp = C.c_void_p
    tvh.tvh_find_essential.argtypes = [vp, vp, C.c_int, vp, C.c_double, C.c_double, C.c_int, vp, vp, vp]
    n_frames, n_kp, n_pairs, size = 46, 10000, 1024, (1920, 1080)
    dev = torch.device("cuda", 0)
    ctx = pkg.GmsContext(0)
    sc = synth.make_multi_view_scene(77, n_frames, size=size, n_kp=n_kp, dist=(-0.12, 0.05, 0.001, -0.0007, 0.01))
    table = batch.FrameTable(ctx, sc["frames"], sc["sizes"], device=dev)
    d_desc = distmod.synth_descriptors_device(n_frames, n_kp, "orb", 0.3, dev)
    code = pkg.GMS_DESC_HAMMING256
    d_prep = torch.zeros(max(ctx.bf_prepared_bytes(code, table.total, n_frames), 16), dtype=torch.uint8, device=dev)
    pairs = distmod.pair_table(n_frames, 0, n_pairs, n_kp)
    d_pairs = torch.from_numpy(pairs.view(np.uint8).reshape(-1)).to(dev)
    total = n_pairs * n_kp
    d_matches = torch.zeros((total, 4), dtype=torch.int32, device=dev)
    d_out = torch.zeros((total, 4), dtype=torch.int32, device=dev)
    d_res = torch.zeros((n_pairs, 4), dtype=torch.int32, device=dev)
    d_c1, d_c2 = torch.zeros(2 * total, dtype=torch.float32, device=dev), torch.zeros(2 * total, dtype=torch.float32, device=dev)
    d_mask = torch.zeros(total, dtype=torch.uint8, device=dev)
    d_p3 = torch.zeros(3 * total, dtype=torch.float64, device=dev)
    d_tv = torch.zeros(n_pairs * types.TWO_VIEW_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    cam = types.make_camera(sc["camera"], sc["dist"])
    torch.cuda.synchronize()
    ctx.bf_prepare_device(code, d_desc.data_ptr(), table.d_frame_off.data_ptr(), n_frames, table.total, d_prep.data_ptr())
    ctx.bfmatch_device(code, d_desc.data_ptr(), d_prep.data_ptr(), table.total, table.d_frame_off.data_ptr(), n_frames, d_pairs.data_ptr(), n_pairs, n_kp,
                       d_matches.data_ptr())
    ctx.filter_device(table.d_pts.data_ptr(), table.d_frame_off.data_ptr(), n_frames, d_pairs.data_ptr(), n_pairs, n_kp, d_matches.data_ptr(),
                      d_out.data_ptr(), d_res.data_ptr(), None, True, True, 6.0)
    ctx.synchronize()
    runs = []
    for r in range(6):
        ctx.two_view_batch_device(cam, table.d_kp.data_ptr(), table.d_frame_off.data_ptr(), n_frames, d_pairs.data_ptr(), n_pairs, n_kp,
                                  d_out.data_ptr(), d_res.data_ptr(), d_c1.data_ptr(), d_c2.data_ptr(), d_mask.data_ptr(), d_p3.data_ptr(),
                                  d_tv.data_ptr(), 0.7, 1.0, 1000)
        ctx.synchronize()
        runs.append((d_tv.cpu().numpy().view(types.TWO_VIEW_DTYPE).copy(), d_mask.cpu().numpy().copy()))
    for r in range(1, 6):
        same = runs[r][0].tobytes() == runs[0][0].tobytes() and np.array_equal(runs[r][1], runs[0][1])
        diff = np.flatnonzero([runs[r][0][i].tobytes() != runs[0][0][i].tobytes() for i in range(n_pairs)])
        print("run", r, "identical to run 0:", same, "pairs that differ:", diff[:10].tolist(), len(diff))
    tv = runs[0][0]
    c1, c2 = d_c1.cpu().numpy().reshape(-1, 2), d_c2.cpu().numpy().reshape(-1, 2)
    camera = np.array(sc["camera"])
    bad = 0
    for i in range(n_pairs):
        k, o = int(tv["n_points"][i]), i * n_kp
        Eh, mh, ith = np.zeros(9), np.zeros(max(k, 1), dtype=np.uint8), C.c_int(0)
        u1, u2 = np.ascontiguousarray(c1[o:o + k]), np.ascontiguousarray(c2[o:o + k])
        good = tvh.tvh_find_essential(u1.ctypes.data, u2.ctypes.data, k, camera.ctypes.data, 0.7, 1.0, 1000, Eh.ctypes.data, mh.ctypes.data, C.byref(ith))
        d = np.abs(Eh.reshape(3, 3) - tv["E"][i]).max()
        if good != int(tv["n_ransac"][i]) or ith.value != int(tv["ransac_iters"][i]) or d > 1e-9:
            bad += 1
            if bad <= 8:
                print("pair", i, "k", k, "host count/iters", good, ith.value, "gpu", int(tv["n_ransac"][i]), int(tv["ransac_iters"][i]), "E diff", d)
    print("pairs where GPU and host build differ:", bad, "of", n_pairs)


if __name__ == "__main__":
    main()
