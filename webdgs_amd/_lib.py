"""ctypes binding of ``libwebdgs_hip.so`` (``include/webdgs.h``).

The library is the product: there is no CPU fallback.  If it is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WDGS_LIB_PATH: another build of the same library (same-box A/B measurements of two source states); never a different implementation
LIB_PATH = os.environ.get("WDGS_LIB_PATH") or os.path.join(_HERE, "lib", "libwebdgs_hip.so")

WDGS_OK, WDGS_E_INVALID, WDGS_E_HIP, WDGS_E_CAPACITY, WDGS_E_STATE = 0, -1, -2, -3, -4


class WdgsError(RuntimeError):
    """Raised for every non-zero return code (the reference throws ``Error``)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[wdgs {code}] {message}")
        self.code = code


class CapacityError(WdgsError):
    pass


class StateError(WdgsError):
    pass


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint32), ("total_ms", C.c_float)]


class TiledForwardConfig(C.Structure):
    _fields_ = [("num_points", C.c_uint32), ("sh_deg", C.c_uint32), ("viewport_width", C.c_uint32), ("viewport_height", C.c_uint32),
                ("gaussian_scale", C.c_float), ("point_size_px", C.c_float), ("max_splat_radius_px", C.c_float), ("render_mode", C.c_uint32),
                ("max_tile_entries", C.c_uint32), ("compat_caps", C.c_uint32)]


class TiledForwardResources(C.Structure):
    _fields_ = [("splat_buffer", C.c_void_p), ("depths_buffer", C.c_void_p), ("tile_keys_buffer", C.c_void_p), ("tile_indices_buffer", C.c_void_p),
                ("tile_offsets_buffer", C.c_void_p), ("tile_counts_buffer", C.c_void_p), ("stats_buffer", C.c_void_p),
                ("num_tiles_x", C.c_uint32), ("num_tiles_y", C.c_uint32), ("total_tiles", C.c_uint32), ("max_tile_entries", C.c_uint32),
                ("settings", C.c_float * 7)]


class TrainingConfig(C.Structure):
    _fields_ = [("lambda_l1", C.c_float), ("lambda_l2", C.c_float), ("lambda_dssim", C.c_float), ("c1", C.c_float), ("c2", C.c_float)]


class TiledBackwardConfig(C.Structure):
    _fields_ = [("num_points", C.c_uint32), ("sh_deg", C.c_uint32), ("viewport_width", C.c_uint32), ("viewport_height", C.c_uint32),
                ("training", TrainingConfig), ("gaussian_scale", C.c_float), ("point_size_px", C.c_float), ("max_splat_radius_px", C.c_float)]


class TiledBackwardResources(C.Structure):
    _fields_ = [("splat_buffer", C.c_void_p), ("tile_offsets_buffer", C.c_void_p), ("tile_indices_buffer", C.c_void_p), ("camera_buffer", C.c_void_p),
                ("alpha_texture", C.c_void_p), ("n_contrib_texture", C.c_void_p)]


class ViewAccumulate(C.Structure):
    _fields_ = [("sums", C.c_void_p), ("visible", C.c_void_p), ("tile_counts", C.c_void_p), ("guard", C.c_void_p), ("overflow_word", C.c_void_p),
                ("first", C.c_int)]


class AdamHyperparameters(C.Structure):
    _fields_ = [("lr_pos", C.c_float), ("lr_color", C.c_float), ("lr_opacity", C.c_float), ("lr_scale", C.c_float), ("lr_rot", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float)]


class OptimizerState(C.Structure):
    _fields_ = [("opt_pos", C.c_void_p), ("opt_rot", C.c_void_p), ("opt_scale", C.c_void_p), ("opt_opacity", C.c_void_p), ("param_sh", C.c_void_p),
                ("state_sh", C.c_void_p)]


class DensifyConfig(C.Structure):
    _fields_ = [("num_views", C.c_uint32), ("clone_threshold", C.c_uint32), ("split_threshold", C.c_float), ("prune_threshold", C.c_float),
                ("max_new_points_per_step", C.c_uint32), ("max_buffer_bytes", C.c_uint64)]


class DensifyPrepared(C.Structure):
    _fields_ = [("action_buffer", C.c_void_p), ("out_count_buffer", C.c_void_p), ("out_offset_buffer", C.c_void_p), ("out_total_buffer", C.c_void_p),
                ("max_out_points", C.c_uint32)]


_P = C.c_void_p
_U = C.c_uint32
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t
DoneCallback = C.CFUNCTYPE(None, C.c_void_p)  # wdgs_done_callback

# name -> (restype, argtypes).  Every symbol include/webdgs.h declares is listed (tests/test_abi.py checks both ways).
SIGNATURES = {
    "wdgs_last_error": (C.c_char_p, []),
    "wdgs_abi_version": (_I, []),
    "wdgs_device_create": (_I, [_I, _P, C.POINTER(_P)]),
    "wdgs_device_destroy": (_I, [_P]),
    "wdgs_device_synchronize": (_I, [_P]),
    "wdgs_device_memory_info": (_I, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "wdgs_device_set_profiling": (_I, [_P, _I]),
    "wdgs_device_get_kernel_times": (_I, [_P, C.POINTER(KernelTime), _U, C.POINTER(_U)]),
    "wdgs_device_reset_kernel_times": (_I, [_P]),
    "wdgs_queue_mark": (_I, [_P, C.POINTER(C.c_uint64)]),
    "wdgs_queue_wait": (_I, [_P, C.c_uint64]),
    "wdgs_device_select_lane": (_I, [_P, _I]),
    "wdgs_device_lane_order": (_I, [_P, _I, _I]),
    "wdgs_device_lane_mark": (_I, [_P, _I, _I]),
    "wdgs_device_lane_wait_mark": (_I, [_P, _I, _I]),
    "wdgs_encoder_begin": (_I, [_P]),
    "wdgs_encoder_finish": (_I, [_P, C.POINTER(_P)]),
    "wdgs_encoder_abort": (_I, [_P]),
    "wdgs_queue_submit": (_I, [_P, _P]),
    "wdgs_command_buffer_destroy": (_I, [_P]),
    "wdgs_queue_on_done": (_I, [_P, DoneCallback, _P]),
    "wdgs_buffer_read_async": (_I, [_P, _P, _Z, _P, _Z]),
    "wdgs_host_alloc": (_I, [_Z, C.POINTER(_P)]),
    "wdgs_host_free": (_I, [_P]),
    "wdgs_tiled_rasterizer_blit": (_I, [_P, _P, _U, _U]),
    "wdgs_densify_prune_encode_decision": (_I, [_P, _U, _P, _P]),
    "wdgs_densify_prune_encode_prefix_sum": (_I, [_P, _U]),
    "wdgs_densify_prune_encode_cap_to_max": (_I, [_P, _U, _U]),
    "wdgs_densify_prune_encode_total_out": (_I, [_P, _U]),
    "wdgs_densify_prune_compute_max_out_points": (_I, [_P, _U, C.POINTER(_U)]),
    "wdgs_densify_prune_get_buffers": (_I, [_P, C.POINTER(DensifyPrepared)]),
    "wdgs_debug_eval_math": (_I, [_P, _U, _U, _P, _P]),
    "wdgs_comm_get_unique_id": (_I, [C.POINTER(C.c_uint8)]),
    "wdgs_comm_create": (_I, [_P, C.POINTER(C.c_uint8), _I, _I, C.POINTER(_P)]),
    "wdgs_comm_destroy": (_I, [_P]),
    "wdgs_comm_world_size": (_I, [_P]),
    "wdgs_comm_rank": (_I, [_P]),
    "wdgs_comm_allreduce_gradients": (_I, [_P, _P, _P, _U]),
    "wdgs_comm_allreduce_counts": (_I, [_P, _P, _U]),
    "wdgs_comm_exchange_gradients": (_I, [_P, _P, _P, _P, _U]),
    "wdgs_comm_allgather_rows": (_I, [_P, _P, _U]),
    "wdgs_comm_broadcast": (_I, [_P, _P, _Z, _I]),
    "wdgs_comm_group_start": (_I, []),
    "wdgs_comm_group_end": (_I, []),
    "wdgs_optimizer_step_f32_range": (_I, [_P, _P, _P, _P, _P, _U, _U, _P]),
    "wdgs_apply_repacked_rows": (_I, [_P, _U, _P, _U, _U, _P, _P, _P]),
    "wdgs_optimizer_set_guard": (_I, [_P, _P]),
    "wdgs_optimizer_set_deferred_sh": (_I, [_P, _P, _I]),
    "wdgs_optimizer_dc_words": (_P, [_P]),
    "wdgs_optimizer_flush_sh": (_I, [_P, _P]),
    "wdgs_optimizer_apply_repacked_rows": (_I, [_P, _P, _U, _U, _P, _P, _P]),
    "wdgs_tiled_forward_set_dc_source": (_I, [_P, _P]),
    "wdgs_tiled_forward_project_views": (_I, [_P, _P, _U, _P, _P]),
    "wdgs_tiled_forward_encode_projected": (_I, [_P]),
    "wdgs_tiled_forward_is_projected": (_I, [_P]),
    "wdgs_tiled_backward_encode_geometry_views": (_I, [_P, _P, _P, _P, _U, _P, _P, _P, _P, _I, _I]),
    "wdgs_guard_accumulate": (_I, [_P, _P, _P, _I]),
    "wdgs_optimizer_state_changed": (_I, [_P]),
    "wdgs_copy_to_host": (_I, [_P, _P, _P, _Z]),
    "wdgs_copy_to_device": (_I, [_P, _P, _P, _Z]),
    "wdgs_memset": (_I, [_P, _P, _I, _Z]),
    "wdgs_copy_buffer_to_buffer": (_I, [_P, _P, _P, _Z]),
    "wdgs_buffer_create": (_I, [_P, _Z, C.POINTER(_P)]),
    "wdgs_buffer_destroy": (_I, [_P]),
    "wdgs_buffer_ptr": (_P, [_P]),
    "wdgs_buffer_size": (_Z, [_P]),
    "wdgs_buffer_write": (_I, [_P, _P, _Z, _P, _Z]),
    "wdgs_buffer_read": (_I, [_P, _P, _Z, _P, _Z]),
    "wdgs_prefix_scanner_create": (_I, [_P, _U, C.POINTER(_P)]),
    "wdgs_prefix_scanner_destroy": (_I, [_P]),
    "wdgs_prefix_scanner_input": (_P, [_P]),
    "wdgs_prefix_scanner_output": (_P, [_P]),
    "wdgs_prefix_scanner_set_count": (_I, [_P, _U]),
    "wdgs_prefix_scanner_scan": (_I, [_P]),
    "wdgs_prefix_scanner_scan_ptr": (_I, [_P, _P, _P, _U]),
    "wdgs_sorter_create": (_I, [_P, _U, _P, C.POINTER(_P)]),
    "wdgs_sorter_destroy": (_I, [_P]),
    "wdgs_sorter_keys": (_P, [_P, _I]),
    "wdgs_sorter_values": (_P, [_P, _I]),
    "wdgs_sorter_sort": (_I, [_P, _U]),
    "wdgs_sorter_final_out_index": (_I, [_P]),
    "wdgs_sorter_capacity": (_U, [_P]),
    "wdgs_tiled_forward_create": (_I, [_P, C.POINTER(TiledForwardConfig), C.POINTER(_P)]),
    "wdgs_tiled_forward_destroy": (_I, [_P]),
    "wdgs_tiled_forward_resize": (_I, [_P, _U]),
    "wdgs_tiled_forward_encode": (_I, [_P, _P, _P, _P, _I]),
    "wdgs_tiled_forward_set_viewport": (_I, [_P, _U, _U]),
    "wdgs_tiled_forward_set_render_mode": (_I, [_P, _U]),
    "wdgs_tiled_forward_set_point_size": (_I, [_P, _F]),
    "wdgs_tiled_forward_set_gaussian_scale": (_I, [_P, _F]),
    "wdgs_tiled_forward_get_resources": (_I, [_P, C.POINTER(TiledForwardResources)]),
    "wdgs_tiled_forward_check": (_I, [_P, C.POINTER(_U)]),
    "wdgs_tiled_forward_set_long_lists": (_I, [_P, _U, _U, _U]),
    "wdgs_tiled_forward_long_list_stats": (_I, [_P, C.POINTER(_U)]),
    "wdgs_tiled_rasterizer_create": (_I, [_P, _P, _U, C.POINTER(_P)]),
    "wdgs_tiled_rasterizer_destroy": (_I, [_P]),
    "wdgs_tiled_rasterizer_encode": (_I, [_P, _U, _U]),
    "wdgs_tiled_rasterizer_get_output": (_I, [_P, C.POINTER(_P)]),
    "wdgs_tiled_rasterizer_get_alpha": (_I, [_P, C.POINTER(_P)]),
    "wdgs_tiled_rasterizer_get_n_contrib": (_I, [_P, C.POINTER(_P)]),
    "wdgs_tiled_rasterizer_get_tile_offsets": (_I, [_P, C.POINTER(_P)]),
    "wdgs_tiled_backward_create": (_I, [_P, C.POINTER(TiledBackwardConfig), C.POINTER(_P)]),
    "wdgs_tiled_backward_resize": (_I, [_P, _U]),
    "wdgs_tiled_backward_destroy": (_I, [_P]),
    "wdgs_tiled_backward_encode": (_I, [_P, _P, _P, C.POINTER(TiledBackwardResources), _P]),
    "wdgs_tiled_backward_encode_raster": (_I, [_P, _P, _P, C.POINTER(TiledBackwardResources)]),
    "wdgs_tiled_backward_encode_geometry": (_I, [_P, _P, _P, C.POINTER(ViewAccumulate)]),
    "wdgs_tiled_backward_compute_loss_only": (_I, [_P, _P, _P]),
    "wdgs_tiled_backward_compute_metric_map": (_I, [_P, _P, _P, _F]),
    "wdgs_tiled_backward_compute_metric_counts": (_I, [_P, C.POINTER(TiledBackwardResources), _U, _I]),
    "wdgs_tiled_backward_normalize_metric_counts": (_I, [_P, _U]),
    "wdgs_tiled_backward_set_viewport": (_I, [_P, _U, _U]),
    "wdgs_tiled_backward_set_training_config": (_I, [_P, C.POINTER(TrainingConfig)]),
    "wdgs_tiled_backward_gradients": (_P, [_P]),
    "wdgs_tiled_backward_metric_counts": (_P, [_P]),
    "wdgs_tiled_backward_set_metric_counts_target": (_I, [_P, _P]),
    "wdgs_tiled_backward_loss_image": (_P, [_P]),
    "wdgs_tiled_backward_metric_map": (_P, [_P]),
    "wdgs_tiled_backward_accumulators": (_P, [_P]),
    "wdgs_tiled_backward_set_gradient_output": (_I, [_P, _I]),
    "wdgs_tiled_backward_metric_minmax": (_P, [_P]),
    "wdgs_downsample_rgba8": (_I, [_P, _P, _U, _U, _P, _U, _U]),
    "wdgs_image_sse_rgb8": (_I, [_P, _P, _P, _U, _P]),
    "wdgs_optimizer_state_sizes": (_I, [_U, C.POINTER(_Z * 6)]),
    "wdgs_optimizer_create": (_I, [_P, _U, C.POINTER(AdamHyperparameters), _P, _P, C.POINTER(OptimizerState), _I, _U, C.POINTER(_P)]),
    "wdgs_optimizer_destroy": (_I, [_P]),
    "wdgs_optimizer_init_from_point_cloud": (_I, [_P, _P, _P]),
    "wdgs_optimizer_step": (_I, [_P, _P, _P, _P, _P]),
    "wdgs_optimizer_step_with_geometry": (_I, [_P, _P, _P, _P, _P, _P]),
    "wdgs_optimizer_step_f32": (_I, [_P, _P, _P, _P, _P]),
    "wdgs_accumulate_gradients": (_I, [_P, _U, _P, _P, _P, _P]),
    "wdgs_store_gradients": (_I, [_P, _U, _P, _P, _P, _P]),
    "wdgs_optimizer_get_iteration": (_U, [_P]),
    "wdgs_optimizer_advance_iteration": (_I, [_P, _U]),
    "wdgs_optimizer_get_hyperparameters": (_I, [_P, C.POINTER(AdamHyperparameters)]),
    "wdgs_optimizer_set_hyperparameters": (_I, [_P, C.POINTER(AdamHyperparameters)]),
    "wdgs_optimizer_get_state": (_I, [_P, C.POINTER(OptimizerState)]),
    "wdgs_optimizer_release_state": (_I, [_P, C.POINTER(OptimizerState)]),
    "wdgs_densify_prune_create": (_I, [_P, C.POINTER(DensifyConfig), C.POINTER(_P)]),
    "wdgs_densify_prune_destroy": (_I, [_P]),
    "wdgs_densify_prune_set_config": (_I, [_P, C.POINTER(DensifyConfig)]),
    "wdgs_densify_prune_ensure_size": (_I, [_P, _U]),
    "wdgs_densify_prune_encode_prepare": (_I, [_P, _U, _P, _P, C.POINTER(DensifyPrepared)]),
    "wdgs_densify_prune_read_total": (_I, [_P, C.POINTER(_U)]),
    "wdgs_densify_prune_encode_scatter": (_I, [_P, _U, _P, _P, C.POINTER(OptimizerState), _U, _I, _P, _P, C.POINTER(OptimizerState)]),
}

_lib = None


def load() -> C.CDLL:
    """Loads the HIP library or raises -- the product path never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C webdgs_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code == WDGS_OK:
        return
    msg = load().wdgs_last_error().decode("utf-8", "replace")
    if code == WDGS_E_CAPACITY:
        raise CapacityError(code, msg)
    if code == WDGS_E_STATE:
        raise StateError(code, msg)
    raise WdgsError(code, msg)
