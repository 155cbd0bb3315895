"""ctypes binding of libdfe.so (C ABI: include/dfe.h).  No CPU fallback: if the shared library
is missing or no gfx950 device is present, every entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFE_LIB") or os.path.join(_HERE, "libdfe.so")  # DFE_LIB: tuning builds only

DFE_OK = 0
DFE_MAX_RATIOS = 10

# keys of dfe_set_option / dfe_get_option (include/dfe.h)
OPTION_KEYS = ("cascade_px", "fine_fuse", "mid_fuse", "fine_nq", "mid_nq", "prep_tiles", "xpose", "xpose_nt", "soft_epilogue", "conv_batch", "conv_nt10",
               "fm64", "fm_rows", "sweep_ovh", "sweep_blocks", "debug_arena", "fm_flat", "fm_split", "conv_narrow", "conv_mfma", "fm_mfma", "arena_contig", "graphs")

c_f32p = C.POINTER(C.c_float)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int)

class FilterLayer(C.Structure):
    """dfe_filter_layer (include/dfe.h)"""
    _fields_ = [("nIn", C.c_int), ("nOut", C.c_int), ("kH", C.c_int), ("kW", C.c_int), ("weight", C.c_void_p), ("bias", C.c_void_p),
                ("conn", C.c_void_p), ("nConn", C.c_int), ("tanh_after", C.c_int)]


# name -> (restype, argtypes); must list every symbol include/dfe.h declares
PROTOTYPES = {
    "dfe_version": (C.c_int, []),
    "dfe_kernel_revision": (C.c_char_p, []),
    "dfe_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "dfe_ctx_destroy": (None, [C.c_void_p]),
    "dfe_last_error": (C.c_char_p, [C.c_void_p]),
    "dfe_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "dfe_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "dfe_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dfe_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfe_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dfe_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dfe_host_register": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dfe_host_unregister": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfe_ego_motion_from_points_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_uint,
                                                C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "dfe_ego_motion_from_flow_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int, C.c_double, C.c_int,
                                              C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "dfe_stage_timers_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "dfe_stage_timers_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "dfe_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dfe_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfe_set_cost_volume_kernel": (C.c_int, [C.c_void_p, C.c_int]),
    "dfe_set_cost_volume_tile": (C.c_int, [C.c_void_p, C.c_int]),
    "dfe_version2_flow_pair_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float,
                                             C.POINTER(FilterLayer), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dfe_spatial_matching_argmin_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 3),
    "dfe_flow_pair_filtered_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(FilterLayer), C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dfe_spatial_matching_strided_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "dfe_spatial_convolution_tanh_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "dfe_ingest_submit_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int)]),
    "dfe_flow_depth_pair_u8_slot": (C.c_int, [C.c_void_p, C.c_int] + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_double, C.c_float] + [C.c_void_p] * 4),
    "dfe_u8_to_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "dfe_min_dim0_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "dfe_rgb2y_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dfe_flow_depth_pair_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_double, C.c_float] + [C.c_void_p] * 4),
    "dfe_multiscale_flow_pair_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [c_i32p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "dfe_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "dfe_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "dfe_last_kernel": (C.c_char_p, [C.c_void_p]),
    "dfe_ssd_cost_volume_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "dfe_ssd_cost_volume_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_float, C.c_void_p]),
    "dfe_flow_depth_pair_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 5),
    "dfe_spatial_matching_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "dfe_radial_matching_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "dfe_argbest_center": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dfe_extract_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p]),
    "dfe_extract_output_marginalized": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    ),
    "dfe_x2yx": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dfe_x2yx_multi": (
        C.c_int,
        [C.c_void_p, C.c_int, C.c_int, c_i32p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int],
    ),
    "dfe_yx2x_multi": (C.c_int64, [C.c_int, C.c_int, c_i32p, C.c_int, C.c_double, C.c_double]),
    "dfe_x2yx_multi_number": (C.c_int, [C.c_int, C.c_int, c_i32p, C.c_int, C.c_int64, c_i64p, c_i64p]),
    "dfe_multi_nclasses": (C.c_int64, [C.c_int, C.c_int, c_i32p, C.c_int]),
    "dfe_ssd_flow_f32": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_double] + [C.c_void_p] * 6,
    ),
    "dfe_set_scratch_limit": (C.c_int, [C.c_void_p, C.c_size_t]),
    "dfe_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "dfe_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfe_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "dfe_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "dfe_profile_read_each": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int]),
    "dfe_flow_tail": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 4,
    ),
    "dfe_flow_to_depth_cartesian": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "dfe_flow_depth_pair_f32": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_double] + [C.c_void_p] * 4,
    ),
    "dfe_downsample_box_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "dfe_pyramid_scale_volume_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "dfe_softmin_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dfe_cascading_add_f32": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_i32p, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dfe_spatial_convolution_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dfe_spatial_convolution_map_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dfe_tanh_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dfe_contrastive_normalization_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "dfe_spatial_convolution_mfma_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "dfe_spatial_convolution_grad_input_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "dfe_spatial_convolution_acc_grad_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_float, C.c_void_p, C.c_void_p]),
    "dfe_spatial_convolution_map_grad_input_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "dfe_spatial_convolution_map_acc_grad_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_float, C.c_void_p, C.c_void_p]),
    "dfe_tanh_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dfe_log2_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "dfe_log2_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dfe_log_softmax_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dfe_log_softmax_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dfe_softmax_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dfe_spatial_matching_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dfe_radial_matching_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dfe_cascade_flow_f32": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dfe_multiscale_flow_pair_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, C.c_int, C.c_void_p, C.c_void_p]),
    "dfe_multiscale_flow_pair_filtered_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, C.c_int,
                                                       C.POINTER(FilterLayer), C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "dfe_multiscale_flow_pair_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "dfe_cascading_add_backward_f32": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_i32p, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dfe_cascade_ring_f32": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), c_i32p] + [C.c_int] * 5 + [C.c_void_p]),
    "dfe_polar_grid_c2p_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "dfe_polar_grid_p2c_f32": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_float] * 4 + [C.c_void_p]),
    "dfe_warp_bilinear_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dfe_flow_to_depth_radial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "dfe_marginal_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "dfe_flow_to_depth_ardrone": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "dfe_postprocess_image_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "dfe_enlarge_mask_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4),
    "dfe_output_extractor_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dfe_epipole": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]),
    "dfe_remove_ego_motion_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_void_p, C.c_void_p]),
    "dfe_undistort_image_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "dfe_foe_from_flow_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dfe_radial_match_argmin_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_int]),
    "dfe_radial_out_shape": (C.c_int, [C.c_void_p, c_i32p, c_i32p, c_i32p]),
    "dfe_radial_flow_depth_pair_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double] + [C.c_void_p] * 9),
}


class RadialParams(C.Structure):
    """dfe_radial_params (include/dfe.h)"""
    _fields_ = [("C", C.c_int), ("hImg", C.c_int), ("wImg", C.c_int), ("hInput", C.c_int), ("wInput", C.c_int), ("hWin", C.c_int),
                ("n1", C.c_int), ("kW1", C.c_int), ("n2", C.c_int), ("kH2", C.c_int), ("tanh_between", C.c_int),
                ("alpha_polar", C.c_float), ("kinfty", C.c_double), ("zero_last_row", C.c_int)]

_lib = None


class DfeError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libdfe error %d: %s" % (code, text))
        self.code = code


def lib():
    """Load libdfe.so (built in-tree by depth-estimation_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libdfe.so not found at %s -- build it with `make -C depth-estimation_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH
            )
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def ratios_array(ratios):
    r = list(int(v) for v in ratios)
    return (C.c_int * len(r))(*r), len(r)
