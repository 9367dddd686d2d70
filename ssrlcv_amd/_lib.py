"""ctypes loader for the HIP C-ABI library.  Fails loudly when the extension has not been built.

Two flavours are built from the same sources (csrc/Makefile): libssrlcv_hip_release.so -- what INTEGRATION.md tells a
deployer to link: every SSRLCV_* developer switch compiled out (csrc/dev_switch.h) -- and libssrlcv_hip.so, the developer
build whose code paths the environment can steer.  The RELEASE flavour is the default here, so bench.py, smoke() and the
parity tests run the library that ships; SSRLCV_DEV_BUILD=1 selects the developer build (the formulation-by-formulation
tests do, in child processes: tests/helpers.py dev_env), SSRLCV_HIP_LIB=<path> any other build (A/B experiments)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
RELEASE_LIB_PATH = os.path.join(_HERE, "libssrlcv_hip_release.so")
DEV_LIB_PATH = os.path.join(_HERE, "libssrlcv_hip.so")
LIB_PATH = os.environ.get("SSRLCV_HIP_LIB") or (DEV_LIB_PATH if os.environ.get("SSRLCV_DEV_BUILD") else RELEASE_LIB_PATH)


def flavour():
    """'release', 'developer' or the path of a custom build: which library this process loads."""
    real = os.path.realpath(LIB_PATH)
    if real == os.path.realpath(RELEASE_LIB_PATH):
        return "release"
    if real == os.path.realpath(DEV_LIB_PATH):
        return "developer"
    return LIB_PATH

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


class HipAbiMismatch(RuntimeError):
    pass


# the SSRLCV_HIP_ABI_VERSION of include/ssrlcv_hip.h this package's ctypes signatures (capi.py) were written against
ABI_VERSION = 3


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C ssrlcv_amd/csrc all release` (there is no CPU fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        # refuse a library of another ABI before the first call into it (a build from before the entry point existed has
        # no ssrlcv_hip_abi_version at all)
        got = lib.ssrlcv_hip_abi_version() if hasattr(lib, "ssrlcv_hip_abi_version") else 1
        if got != ABI_VERSION:
            raise HipAbiMismatch("%s has ABI version %d, this package binds version %d (include/ssrlcv_hip.h "
                                 "SSRLCV_HIP_ABI_VERSION): rebuild the library" % (LIB_PATH, got, ABI_VERSION))
        _lib = lib
        _lib.ssrlcv_hip_version.restype = ctypes.c_char_p
        _lib.ssrlcv_hip_status_string.restype = ctypes.c_char_p
        for name in ("ssrlcv_hip_match_workspace_bytes", "ssrlcv_sift_plan_workspace_bytes",
                     "ssrlcv_hip_ba_sweep2_workspace_bytes", "ssrlcv_hip_sort_workspace_bytes",
                     "ssrlcv_hip_select_pair_workspace_bytes", "ssrlcv_hip_filter_workspace_bytes", "ssrlcv_hip_merge_workspace_bytes"):
            getattr(_lib, name).restype = ctypes.c_size_t
        _lib.ssrlcv_sift_plan_max_features.restype = ctypes.c_uint32
    return _lib


# every symbol include/ssrlcv_hip.h declares (checked by tests/test_capi_symbols.py without a GPU)
EXPORTED = [
    "ssrlcv_hip_abi_version", "ssrlcv_hip_version", "ssrlcv_hip_status_string",
    "ssrlcv_hip_device_count", "ssrlcv_hip_malloc", "ssrlcv_hip_free", "ssrlcv_hip_host_malloc",
    "ssrlcv_hip_host_free", "ssrlcv_hip_memcpy", "ssrlcv_hip_memset", "ssrlcv_hip_device_synchronize",
    "ssrlcv_hip_generate_bundles", "ssrlcv_hip_generate_pushbroom_bundles", "ssrlcv_hip_triangulate2",
    "ssrlcv_hip_triangulateN", "ssrlcv_hip_ba_sweep2_workspace_bytes", "ssrlcv_hip_ba_sweep2",
    "ssrlcv_hip_pose_lm_terms", "ssrlcv_hip_pose_cost",
    "ssrlcv_projection_matrix_host", "ssrlcv_hip_match_workspace_bytes", "ssrlcv_hip_set_match_arithmetic", "ssrlcv_hip_get_match_arithmetic", "ssrlcv_hip_seed_distances_u8x128",
    "ssrlcv_hip_match_u8x128", "ssrlcv_hip_compact_matches", "ssrlcv_hip_compact_matches_async", "ssrlcv_hip_keypoints_from_members",
    "ssrlcv_hip_matchset_from_matches", "ssrlcv_merge_matches_host", "ssrlcv_merge_matches_host_mode", "ssrlcv_host_free", "ssrlcv_assign_pairs_host",
    "ssrlcv_hip_merge_workspace_bytes", "ssrlcv_hip_merge_matches",
    "ssrlcv_hip_sort_workspace_bytes", "ssrlcv_hip_sort_keys_u32",
    "ssrlcv_hip_error_sample_cutoff", "ssrlcv_hip_filter_workspace_bytes", "ssrlcv_hip_filter_matchset",
    "ssrlcv_hip_select_pair_workspace_bytes", "ssrlcv_hip_select_pair_bundles",
    "ssrlcv_hip_convert_to_bw", "ssrlcv_hip_u8_to_f32", "ssrlcv_hip_upsample2x", "ssrlcv_hip_upsample2x_u8", "ssrlcv_hip_bin2x",
    "ssrlcv_gauss_kernel_host", "ssrlcv_hip_gauss_sep_conv", "ssrlcv_hip_minmax", "ssrlcv_hip_normalize",
    "ssrlcv_hip_dog_normalised_sub",
    "ssrlcv_sift_plan_create", "ssrlcv_sift_plan_destroy", "ssrlcv_sift_plan_workspace_bytes",
    "ssrlcv_sift_plan_max_features", "ssrlcv_hip_sift_build_dog", "ssrlcv_hip_sift_describe",
    "ssrlcv_hip_sift_extract", "ssrlcv_hip_sift_stage", "ssrlcv_sift_plan_level", "ssrlcv_sift_plan_keypoints",
    "ssrlcv_sift_plan_set_stop_stage", "ssrlcv_sift_plan_set_stage_event", "ssrlcv_hip_math_eval", "ssrlcv_sift_plan_overflow",
    "ssrlcv_hip_find_extrema", "ssrlcv_hip_compact_workspace_bytes", "ssrlcv_hip_compact_addresses", "ssrlcv_hip_compact_thetas",
    "ssrlcv_hip_compact_keypoints", "ssrlcv_hip_fill_extrema", "ssrlcv_hip_flag_noise", "ssrlcv_hip_refine_location",
    "ssrlcv_hip_flag_edges", "ssrlcv_hip_check_keypoints", "ssrlcv_hip_pixel_gradients", "ssrlcv_hip_compute_thetas",
    "ssrlcv_hip_expand_keypoints", "ssrlcv_hip_fill_descriptors",
]
