"""Load a tests/golden/*.npz fixture back into the structured arrays the API takes."""
import glob
import importlib
import os

import numpy as np

synth = importlib.import_module("sfm-gms_amd.synth")
types = importlib.import_module("sfm-gms_amd.types")
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
                  if not n.startswith(("refdll_", "image_")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = np.zeros(len(z["query"]), dtype=types.DMATCH_DTYPE)
    m["queryIdx"], m["trainIdx"], m["imgIdx"], m["distance"] = z["query"], z["train"], z["img_idx"], z["distance"]
    c = dict(size1=tuple(int(v) for v in z["size1"]), size2=tuple(int(v) for v in z["size2"]),
             kp1=synth.make_keypoints(z["xy1"]), kp2=synth.make_keypoints(z["xy2"]), matches=m)
    want = {}
    for rot in (0, 1):
        for scale in (0, 1):
            tag = f"r{rot}s{scale}"
            mask = np.unpackbits(z["mask_" + tag])[: len(m)].astype(bool)
            want[(bool(rot), bool(scale))] = (mask, tuple(int(v) for v in z["best_" + tag]))
    return c, want
