"""MI355X-native GMS match filter -- host-side mirror of the reference's boundary.

The reference's hot path is one call, ``cv::xfeatures2d::matchGMS`` (call sites
``SfM-GMS/SfM-GMS/FeatureMatchUtil.cpp:69`` and ``DisparityUtil.cpp:149,299``). This package is the
Python face of the C ABI in ``include/gms.h``; every compute call goes through
``csrc/libgms_hip.so`` (hand-written HIP for gfx950). There is no CPU fallback: importing works
without a GPU, computing does not.

The directory name has a dash (it is fixed by the project layout), so load it with
``importlib.import_module("sfm-gms_amd")``.
"""
from .types import (KEYPOINT_DTYPE, DMATCH_DTYPE, PAIR_DTYPE, RESULT_DTYPE, GmsError,  # noqa: F401
                    GMS_DESC_HAMMING256, GMS_DESC_L2_F32X128, GMS_DETECT_BORDER)
from .capi import load_library, library_path, EXPORTED_SYMBOLS  # noqa: F401
from .api import matchGMS, GmsContext  # noqa: F401
from .sharding import all_pairs_count, pair_from_index, shard_range  # noqa: F401

__all__ = [
    "KEYPOINT_DTYPE", "DMATCH_DTYPE", "PAIR_DTYPE", "RESULT_DTYPE", "GmsError",
    "load_library", "library_path", "EXPORTED_SYMBOLS", "matchGMS", "GmsContext",
    "all_pairs_count", "pair_from_index", "shard_range", "GMS_DESC_HAMMING256", "GMS_DESC_L2_F32X128", "GMS_DETECT_BORDER",
]
