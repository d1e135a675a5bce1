#!/usr/bin/env python3
"""Writes tests/golden/image_stereo_pair_450x375.npz from the reference's own input images (data, not source):
    /root/reference/SfM-GMS/SourceImages/left1.png, right1.png   the 450 x 375 stereo pair of DisparityUtil.cpp (main.cpp's disparity demo)
    /root/reference/SfM-GMS/SourceImages/left_gt1.png            its ground-truth disparity map (DisparityUtil.cpp:179-201 compares against it)
as 8-bit grey arrays: grey = (299 R + 587 G + 114 B + 500) // 1000 (the weights of cv::cvtColor's BGR2GRAY, in integers). The GPU box has
no /root/reference; the tests and tools read this fixture. Run here: python tests/golden/make_image_fixture.py"""
import os
import numpy as np
from PIL import Image

SRC = "/root/reference/SfM-GMS/SourceImages"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image_stereo_pair_450x375.npz")


def grey(name):
    im = np.array(Image.open(os.path.join(SRC, name)).convert("RGB")).astype(np.int64)
    return ((299 * im[..., 0] + 587 * im[..., 1] + 114 * im[..., 2] + 500) // 1000).astype(np.uint8)


gt = np.array(Image.open(os.path.join(SRC, "left_gt1.png")))
if gt.ndim == 3:
    gt = gt[..., 0]
np.savez_compressed(OUT, left=grey("left1.png"), right=grey("right1.png"), gt=gt.astype(np.uint8))
print(OUT, os.path.getsize(OUT), {k: v.shape for k, v in np.load(OUT).items()})
