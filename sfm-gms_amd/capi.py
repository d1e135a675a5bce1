"""ctypes binding of include/gms.h -> csrc/libgms_hip.so. Fails loudly if the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

# Every symbol include/gms.h declares; tests check the built library exports all of them.
EXPORTED_SYMBOLS = [
    "gms_match", "gms_match_ctx", "gms_ctx_create", "gms_ctx_destroy", "gms_ctx_set_stream",
    "gms_ctx_synchronize", "gms_ctx_reserve", "gms_ctx_query", "gms_ctx_set_option", "gms_frame_table_bytes", "gms_normalize_device", "gms_filter_device",
    "gms_filter_host_batch", "gms_bf_prepared_bytes", "gms_bf_prepare_device", "gms_bfmatch_device", "gms_disparity_device",
    "gms_gather_points_device", "gms_triangulate_device", "gms_recover_pose_device",
    "gms_gather_points_batch_device", "gms_find_essential_batch_device", "gms_recover_pose_batch_device", "gms_triangulate_batch_device",
    "gms_two_view_batch_device", "gms_disparity_batch_device", "gms_dataset_write", "gms_dataset_read", "gms_dataset_free", "gms_max_matches",
    "gms_last_hip_error", "gms_error_string", "gms_version", "gms_selftest_threshold", "gms_selftest_five_point",
    "gms_detect_workspace_bytes", "gms_detect_batch_device", "gms_describe_device",
]

_lib = None


def library_path():
    return os.path.join(_HERE, "csrc", "libgms_hip.so")


def load_library():
    """Load the HIP extension. No fallback: a missing .so is an error (build with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: the HIP extension is not built "
                          f"(run `python -c 'import __graft_entry__ as g; g.build()'`)")
    # torch (used by batch.py for device memory and streams) ships its own copy of the HIP runtime; it
    # must be the first one loaded in a process that uses both, or torch finds "no HIP GPUs". Nothing of
    # torch is called here.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    lib.gms_match.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, i32, i32, dbl, vp, C.POINTER(i32)]
    lib.gms_match_ctx.argtypes = [vp, vp, i32, i32, i32, vp, i32, i32, i32, vp, i32, i32, i32, dbl, vp,
                                  C.POINTER(i32), vp]
    lib.gms_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.gms_ctx_destroy.argtypes = [vp]
    lib.gms_ctx_set_stream.argtypes = [vp, vp]
    lib.gms_ctx_synchronize.argtypes = [vp]
    lib.gms_ctx_reserve.argtypes = [vp, i32, i32, i32, i32]
    lib.gms_ctx_query.argtypes = [vp, i32, C.POINTER(i64)]
    lib.gms_ctx_set_option.argtypes = [vp, i32, i32]
    lib.gms_filter_host_batch.argtypes = [vp, vp, vp, vp, i32, vp, i32, vp, i32, i32, dbl, vp, vp]
    lib.gms_bf_prepared_bytes.argtypes = [i32, i64, i32]
    lib.gms_bf_prepare_device.argtypes = [vp, i32, vp, vp, i32, i64, vp]
    lib.gms_bfmatch_device.argtypes = [vp, i32, vp, vp, i64, vp, i32, vp, i32, i32, vp]
    lib.gms_disparity_device.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]
    lib.gms_gather_points_device.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp]
    lib.gms_triangulate_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    lib.gms_recover_pose_device.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    lib.gms_gather_points_batch_device.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.gms_find_essential_batch_device.argtypes = [vp, vp, dbl, dbl, i32, vp, i32, vp, vp, vp, vp]
    lib.gms_recover_pose_batch_device.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp]
    lib.gms_triangulate_batch_device.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.gms_two_view_batch_device.argtypes = [vp, vp, dbl, dbl, i32, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.gms_disparity_batch_device.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, i64, i32, vp, i64, vp, vp]
    lib.gms_normalize_device.argtypes = [vp, vp, vp, vp, i32, i64, vp]
    lib.gms_filter_device.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, i32, i32, dbl, vp, vp, vp]
    lib.gms_selftest_threshold.argtypes = [vp, vp, vp, vp, dbl, i32, vp]
    lib.gms_selftest_five_point.argtypes = [vp, vp, i32, vp, vp]
    lib.gms_detect_workspace_bytes.argtypes = [i32, i32, i32, i32]
    lib.gms_detect_workspace_bytes.restype = C.c_size_t
    lib.gms_detect_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, C.c_size_t, vp, vp, vp]
    lib.gms_describe_device.argtypes = [vp, vp, i32, i32, vp, i32, vp, C.c_size_t, vp, vp]
    lib.gms_max_matches.argtypes = []
    lib.gms_last_hip_error.argtypes = []
    lib.gms_error_string.argtypes = [i32]
    lib.gms_error_string.restype = C.c_char_p
    lib.gms_version.argtypes = []
    lib.gms_version.restype = C.c_char_p
    lib.gms_dataset_write.argtypes = [C.c_char_p, vp]
    lib.gms_dataset_read.argtypes = [C.c_char_p, vp]
    lib.gms_dataset_free.argtypes = [vp]
    for name in EXPORTED_SYMBOLS:
        if name not in ("gms_error_string", "gms_version"):
            getattr(lib, name).restype = i32
    lib.gms_bf_prepared_bytes.restype = i64
    lib.gms_dataset_free.restype = None
    lib.gms_frame_table_bytes.argtypes = [i64]
    lib.gms_frame_table_bytes.restype = i64
    _lib = lib
    return lib
