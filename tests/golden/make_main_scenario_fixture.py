#!/usr/bin/env python3
"""Writes tests/golden/image_main_scenario_1080p.npz from the reference's own input images (data, not source): the two photographs its
main() feeds to SIFT_matchGMS (main.cpp:19-20: SourceImages/Disparity_L.jpg, Disparity_R.jpg, 1920 x 1080), as 8-bit grey arrays
    left    Disparity_L.jpg                                      (main.cpp:19)
    right   Disparity_R.jpg                                      (main.cpp:20; the "normal camera change" pair, main.cpp:32)
    right_1000  Disparity_R.jpg resized to 1000 x 1000           (main.cpp:44: resize(img2, scale_img2, Size(1000, 1000)); size1 != size2)
The third pair of main() -- the right image turned by 180 degrees (main.cpp:36) -- is right[::-1, ::-1]: the tests build it themselves.
grey = (299 R + 587 G + 114 B + 500) // 1000 (the weights of cv::cvtColor's BGR2GRAY, in integers). The resize is PIL's bilinear filter on
the colour image (cv::resize's INTER_LINEAR samples differently when shrinking: the fixture is this file's pixels, not OpenCV's).
The GPU box has no /root/reference; tests/test_gpu_main_scenario.py and bench.py read this fixture. Run here:
    python tests/golden/make_main_scenario_fixture.py"""
import os
import numpy as np
from PIL import Image

SRC = "/root/reference/SfM-GMS/SourceImages"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image_main_scenario_1080p.npz")


def grey(im):
    a = np.array(im.convert("RGB")).astype(np.int64)
    return ((299 * a[..., 0] + 587 * a[..., 1] + 114 * a[..., 2] + 500) // 1000).astype(np.uint8)


left = Image.open(os.path.join(SRC, "Disparity_L.jpg"))
right = Image.open(os.path.join(SRC, "Disparity_R.jpg"))
assert left.size == (1920, 1080) and right.size == (1920, 1080)
np.savez_compressed(OUT, left=grey(left), right=grey(right), right_1000=grey(right.convert("RGB").resize((1000, 1000), Image.BILINEAR)))
print(OUT, os.path.getsize(OUT), {k: v.shape for k, v in np.load(OUT).items()})
