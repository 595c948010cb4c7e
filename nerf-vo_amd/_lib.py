"""ctypes binding of libnerfvo_hip.so (C-ABI declared in include/nerfvo_hip.h).

No torch types cross this boundary: tensors are passed as raw device pointers (``Tensor.data_ptr()``)
and the stream as ``torch.cuda.current_stream().cuda_stream``.  The product path has NO fallback:
if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "lib" / "libnerfvo_hip.so"

_lib = None

# name -> (restype, argtypes); mirrors include/nerfvo_hip.h one to one
_u32, _u64, _i64, _int = C.c_uint32, C.c_uint64, C.c_int64, C.c_int
_i32 = C.c_int32
_p = C.c_void_p
_f = C.c_float


class WeightsPdfArgs(C.Structure):
    """mirror of nvo_weights_pdf_args"""
    _fields_ = [("R", _u32), ("S", _u32), ("S_out", _u32), ("pre", _p), ("pre_stride", _u32), ("x01", _p),
                ("sbins", _p), ("tbins", _p), ("density_bias", _f), ("sigma", _p), ("weights", _p),
                ("anneal", _f), ("histogram_padding", _f), ("near_plane", _f), ("far_plane", _f),
                ("jitter", _p), ("sbins_out", _p), ("tbins_out", _p), ("anneal_dev", _p),
                ("origins", _p), ("directions", _p), ("x01_out", _p), ("act_bf16", _int)]


class MainLossArgs(C.Structure):
    """mirror of nvo_main_loss_args"""
    _fields_ = [("R", _u32), ("S", _u32), ("pre", _p), ("pre_stride", _u32), ("rgb", _p), ("rgb_stride", _u32),
                ("x01", _p), ("sbins", _p), ("tbins", _p), ("density_bias", _f), ("gt_rgb", _p),
                ("gt_depth", _p), ("directions_norm", _p), ("rgb_mult", _f), ("distortion_mult", _f),
                ("depth_mult", _f), ("depth_sigma", _f), ("inv_rays", _f), ("depth_level_div", _f),
                ("loss_scale", _f), ("out_rgb", _p), ("out_depth", _p), ("out_expected_depth", _p),
                ("out_accumulation", _p), ("weights", _p), ("losses", _p), ("dpre", _p), ("dpre_stride", _u32),
                ("drgb", _p), ("drgb_stride", _u32),
                ("dsigma_dx", _p), ("dsigma_inv_scale", _f), ("gt_normal", _p), ("normal_mult", _f),
                ("out_normals", _p), ("act_bf16", _int), ("loss_scale_dev", _p), ("nonfinite_flag", _p),
                ("tile_live", _p)]


class PropLossArgs(C.Structure):
    """mirror of nvo_prop_loss_args"""
    _fields_ = [("R", _u32), ("S", _u32), ("S_main", _u32), ("pre", _p), ("pre_stride", _u32), ("x01", _p),
                ("sbins", _p), ("tbins", _p), ("sbins_main", _p), ("weights_main", _p), ("density_bias", _f),
                ("gt_depth", _p), ("directions_norm", _p), ("interlevel_mult", _f), ("depth_mult", _f),
                ("depth_sigma", _f), ("inv_rays", _f), ("depth_level_div", _f), ("loss_scale", _f),
                ("losses", _p), ("dpre", _p), ("dpre_stride", _u32), ("act_bf16", _int), ("loss_scale_dev", _p),
                ("nonfinite_flag", _p)]


class ColorArgs(C.Structure):
    """mirror of nvo_color_args"""
    _fields_ = [("R", _u32), ("S", _u32), ("sh", _p), ("base_out", _p), ("embedding", _p), ("cam_idx", _p),
                ("weights", _p), ("rgb", _p), ("hidden", _p), ("drgb", _p), ("d_base_out", _p),
                ("d_embedding", _p), ("d_sh", _p), ("d_weights", _p), ("act_bf16", _int), ("det_scratch", _p),
                ("det_scratch_bytes", _u64), ("n_cameras", _u32), ("nonfinite_flag", _p), ("dw_replicas", _p),
                ("n_dw_replicas", _u32), ("tile_live", _p), ("tile_live_count", _p)]


class RayHeadArgs(C.Structure):
    """mirror of nvo_ray_head_args"""
    _fields_ = [("R", _u32), ("S", _u32), ("seed", _u32), ("n_jitter", _u32), ("step_dev", _p), ("extent_dev", _p),
                ("intrinsics", _p), ("c2w", _p), ("c2w_stride", _u32), ("corrections", _p), ("H", _u32), ("W", _u32),
                ("images", _p), ("depths", _p), ("normals", _p), ("near_plane", _f), ("far_plane", _f),
                ("ray_indices", _p), ("jitter", _p), ("origins", _p), ("directions", _p), ("directions_norm", _p),
                ("pixel_area", _p), ("cam_idx", _p), ("gt_rgb", _p), ("gt_depth", _p), ("gt_normal", _p), ("dirs01", _p),
                ("sh", _p), ("sh_bf16", _int), ("sbins", _p), ("tbins", _p), ("x01", _p)]


class FusedAdamArgs(C.Structure):
    """nvo_fused_adam_args (include/nerfvo_hip.h): optimiser step inside a hash grid's parameter backward."""
    _fields_ = [("params", _p), ("params_half", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("hyper_dev", _p),
                ("bias_dev", _p), ("loss_scale_dev", _p), ("skip_flag", _p), ("lr", _f), ("grad_scale", _f),
                ("beta1", _f), ("beta2", _f), ("eps", _f), ("step", _u32), ("ema", _p), ("ema_half", _p), ("ema_decay", _f),
                ("ema_step_dev", _p)]


class AdamGroup(C.Structure):
    """nvo_adam_group (include/nerfvo_hip.h)"""
    _fields_ = [("offset", C.c_uint64), ("n", C.c_uint64), ("lr", C.c_float), ("step", C.c_uint32),
                ("hyper_dev", C.c_void_p), ("bias_dev", C.c_void_p), ("flag_slot", C.c_uint32),
                ("flag_slot_set", C.c_uint32), ("weight_decay", C.c_float), ("weight_decay_set", C.c_uint32)]


class AdamTail(C.Structure):
    """nvo_adam_tail (include/nerfvo_hip.h)"""
    _fields_ = [("ema", C.c_void_p), ("ema_half", C.c_void_p), ("ema_decay", C.c_float), ("ema_step_dev", C.c_void_p),
                ("ema_flag_slot", C.c_uint32), ("ema_commit", C.c_uint32), ("done_counter", C.c_void_p),
                ("n_commit_groups", C.c_uint32), ("active_mask", C.c_uint32), ("scale_mask", C.c_uint32),
                ("applied", C.c_void_p), ("scale", C.c_void_p), ("growth_tracker", C.c_void_p),
                ("growth_factor", C.c_float), ("backoff_factor", C.c_float), ("growth_interval", C.c_uint32),
                ("min_scale", C.c_float), ("max_scale", C.c_float), ("bias", C.c_void_p)]


class DepthAlignArgs(C.Structure):
    """mirror of nvo_depth_align_args"""
    _fields_ = [("K", _u32), ("M", _u32), ("P", _u32), ("H", _u32), ("W", _u32), ("patches", _p), ("noise", _p),
                ("frames_depth", _p), ("out_depth", _p), ("scratch", _p), ("scale_shift_out", _p)]


class NgpRgbArgs(C.Structure):
    """mirror of nvo_ngp_rgb_args"""
    _fields_ = [("capacity", _u32), ("sh", _p), ("density_out", _p), ("ray_idx", _p), ("weights", _p), ("rgb_out", _p),
                ("hidden", _p), ("d_rgb_out", _p), ("d_density_out", _p), ("d_density_pre", _p), ("d_weights", _p),
                ("nonfinite_flag", _p), ("dw_replicas", _p), ("n_dw_replicas", _u32)]


class NgpLossArgs(C.Structure):
    """mirror of nvo_ngp_loss_args"""
    _fields_ = [("R", _u32), ("capacity", _u32), ("counts", _p), ("offsets", _p), ("ray_idx", _p), ("t", _p), ("dt", _p),
                ("density_out", _p), ("density_stride", _u32), ("rgb_out", _p), ("rgb_stride", _u32),
                ("background", _p), ("gt_rgb", _p), ("gt_depth", _p), ("directions_norm", _p), ("rgb_mult", _f),
                ("depth_mult", _f), ("inv_rays", _f), ("loss_scale", _f), ("out_rgb", _p), ("out_depth", _p),
                ("out_accumulation", _p), ("losses", _p), ("d_rgb_out", _p), ("d_rgb_stride", _u32),
                ("d_density_pre", _p), ("carry_in", _p), ("carry_out", _p), ("accumulate_outputs", _u32),
                ("train_min_transmittance", _f), ("gt_depth_cov", _p), ("ray_state", _p), ("R_dev", _p), ("world_size", _u32)]


class NgpAliveArgs(C.Structure):
    """mirror of nvo_ngp_alive_args"""
    _fields_ = [("R", _u32), ("counts", _p), ("offsets", _p), ("dt", _p), ("density_out", _p), ("density_stride", _u32),
                ("min_transmittance", _f), ("kept", _p), ("state", _p), ("R_dev", _p), ("resume_in", _p), ("carry_in", _p),
                ("kept_base", _u32), ("t_next", _p), ("resume_out", _p), ("carry_out", _p)]


_SIGNATURES = {
    "nvo_last_error": (C.c_char_p, []),
    "nvo_version": (_int, []),
    "nvo_profile_enable": (_int, [_int]),
    "nvo_profile_summary": (_i64, [C.c_char_p, _u64]),
    "nvo_create_encoding": (_int, [_u32, C.c_char_p, C.POINTER(_p)]),
    "nvo_create_network": (_int, [_u32, _u32, C.c_char_p, C.POINTER(_p)]),
    "nvo_create_network_with_input_encoding": (_int, [_u32, _u32, C.c_char_p, C.c_char_p, C.POINTER(_p)]),
    "nvo_destroy": (_int, [_p]),
    "nvo_n_input_dims": (_u32, [_p]),
    "nvo_n_output_dims": (_u32, [_p]),
    "nvo_padded_output_dims": (_u32, [_p]),
    "nvo_n_params": (_u64, [_p]),
    "nvo_initial_params": (_int, [_p, _u64, _p]),
    "nvo_ctx_bytes": (_u64, [_p, _u32]),
    "nvo_set_option": (_int, [_p, C.c_char_p, _i64]),
    "nvo_fwd": (_int, [_p, _p, _u32, _p, _p, _p, _p]),
    "nvo_bwd": (_int, [_p, _p, _u32, _p, _p, _p, _p, _p, _p, _p]),
    "nvo_bwd_fork": (_int, [_p, _p, _p, _u32, _p, _p, _p, _p, _p, _p, _p]),
    "nvo_grid_describe": (_int, [_p, _p, _p]),
    "nvo_grid_indices": (_int, [_p, _p, _u32, _p, _p]),
    # group B
    "nvo_raygen": (_int, [_p, _u32, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nvo_se3_exp_map": (_int, [_p, _u32, _p, _p]),
    "nvo_pose_exp_map": (_int, [_p, _u32, _p, _p, _int]),
    "nvo_positions_bwd": (_int, [_p, _u32, _u32, _p, _p, _p, _p, _p, _p]),
    "nvo_sh_bwd_input_f32": (_int, [_p, _u32, _u32, _p, _p, _p]),
    "nvo_pose_bwd": (_int, [_p, _u32, _p, _p, _p, _p, _p, _p, _p]),
    "nvo_se3_exp_map_bwd": (_int, [_p, _u32, _p, _p, _f, _f, _f, _p, _p, _int]),
    "nvo_se3_exp_map_bwd_scaled": (_int, [_p, _u32, _p, _p, _f, _f, _f, _p, _p, _int, _p]),
    "nvo_gather_pixels": (_int, [_p, _u32, _p, _u32, _u32, _u32, _p, _p]),
    "nvo_sample_lindisp": (_int, [_p, _u32, _u32, _f, _f, _p, _p, _p]),
    "nvo_sample_pixels": (_int, [_p, _u32, _u32, _p, _p, _p, _p, _u32]),
    "nvo_lindisp_positions": (_int, [_p, _u32, _u32, _f, _f, _p, _p, _p, _p, _p, _p]),
    "nvo_gather_targets": (_int, [_p, _u32, _p, _u32, _u32, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nvo_sample_positions": (_int, [_p, _u32, _u32, _p, _p, _p, _p]),
    "nvo_dirs01": (_int, [_p, _u32, _p, _p]),
    "nvo_sh_encode": (_int, [_p, _u32, _u32, _p, _p]),
    "nvo_sh_encode_t": (_int, [_p, _u32, _u32, _p, _p, _int]),
    "nvo_ray_head": (_int, [_p, C.POINTER(RayHeadArgs)]),
    "nvo_rays_given": (_int, [_p, _u32, _p, _p, _p, _p, _u32, _u32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "nvo_ray_head_zero": (_int, [_p, C.POINTER(RayHeadArgs), _u32, _p, _p]),
    # group C
    "nvo_weights_pdf": (_int, [_p, C.POINTER(WeightsPdfArgs)]),
    "nvo_main_render_loss": (_int, [_p, C.POINTER(MainLossArgs)]),
    "nvo_prop_loss": (_int, [_p, C.POINTER(PropLossArgs)]),
    "nvo_prop_loss_pair": (_int, [_p, _p, _p]),
    # group D
    "nvo_nerfacto_color_fwd": (_int, [_p, C.POINTER(ColorArgs)]),
    "nvo_nerfacto_color_bwd": (_int, [_p, C.POINTER(ColorArgs)]),
    "nvo_color_det_scratch_bytes": (_u64, [_u32, _u32]),
    "nvo_pose_bwd_cams": (_int, [_p, _u32, _p, _p, _p, _p, _p, _p, _p, _u32]),
    "nvo_pose_bwd_det": (_int, [_p, _u32, _p, _p, _p, _p, _p, _p, _p, _p, _u32]),
    # group F
    "nvo_occ_march_scratch_bytes": (_u64, [_u32]),
    "nvo_occ_march": (_int, [_p, _u32, _p, _p, _p, _int, _f, _f, _p, _u32, _p, _p, _p, _p, _p, _p, _u64]),
    "nvo_occ_march_resume": (_int, [_p, _u32, _p, _p, _p, _int, _f, _f, _p, _u32, _p, _p, _p, _p, _p, _p, _u64, _p, _u32, _p]),
    "nvo_occ_update": (_int, [_p, _int, _p, _p, _f, _f, _p, _p]),
    "nvo_occ_cell_positions": (_int, [_p, _int, _p, _p]),
    "nvo_occ_mark_untrained": (_int, [_p, _int, _p, _u32, _p, _p, _u32, _u32, _f]),
    "nvo_occ_sample_cells": (_int, [_p, _u32, _u32, _u32, _u32, _u32, _u32, _int, _p, _f, _f, _f, _p, _p]),
    "nvo_occ_march_runs": (_int, [_p, _u32, _p, _p, _p, _int, _f, _f, _p, _p, _p, _u64, _p, _u32, _p, _p, _u32]),
    "nvo_occ_pack": (_int, [_p, _u32, _p, _u32, _p, _p, _p, _p, _u64, _p, _p, _p, _p, _u32]),
    "nvo_occ_pack_state_bytes": (_u64, []),
    "nvo_occ_pack_fused": (_int, [_p, _u32, _p, _u32, _p, _p, _p, _p, _u64, _p, _p, _p, _p, _u32, _p, _p, _p, _f, _f, _p, _u32]),
    "nvo_ngp_count_alive": (_int, [_p, C.POINTER(NgpAliveArgs)]),
    "nvo_ngp_positions_live": (_int, [_p, _u32, _p, _p, _p, _p, _f, _f, _p, _p]),
    "nvo_ngp_positions_bwd_dev": (_int, [_p, _u32, _u32, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p]),
    "nvo_ngp_positions": (_int, [_p, _u32, _p, _p, _p, _p, _f, _f, _p]),
    "nvo_depth_align_scratch_bytes": (_u64, [_u32, _u32]),
    "nvo_depth_align": (_int, [_p, C.POINTER(DepthAlignArgs)]),
    "nvo_ngp_positions_bwd": (_int, [_p, _u32, _u32, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p]),
    "nvo_ngp_rgb_fwd": (_int, [_p, C.POINTER(NgpRgbArgs)]),
    "nvo_ngp_rgb_bwd": (_int, [_p, C.POINTER(NgpRgbArgs)]),
    "nvo_ngp_composite_loss": (_int, [_p, C.POINTER(NgpLossArgs)]),
    "nvo_ngp_thickness": (_int, [_p, _u32, _p, _u32, _int, _p]),
    "nvo_ngp_thickness_splat": (_int, [_p, _u32, _p, _u32, _p, _p]),
    "nvo_fill_i32": (_int, [_p, _u32, _p, _i32]),
    # group E
    "nvo_adam_step": (_int, [_p, _u64, _p, _p, _p, _int, _p, _p, _f, _f, _f, _f, _u32, _f, _f, _p, _p]),
    "nvo_write_floats": (_int, [_p, _p, _u32, _p]),
    "nvo_nonfinite_flag": (_int, [_p, _u64, _p, _int, _p]),
    "nvo_nonfinite_flag_or": (_int, [_p, _u64, _p, _int, _p]),
    "nvo_adam_step_groups": (_int, [_p, _u32, _p, _p, _p, _p, _int, _p, _p, _f, _f, _f, _f, _f, _p]),
    "nvo_nonfinite_flag_ranges": (_int, [_p, _u32, _p, _p, _p, _int, _p]),
    "nvo_nonfinite_flag_ranges_or": (_int, [_p, _u32, _p, _p, _p, _int, _p]),
    "nvo_nonfinite_flag_spans_or": (_int, [_p, _u32, _p, _p, _p, _p, _int, _p]),
    "nvo_cast_half": (_int, [_p, _u64, _p, _p]),
    "nvo_zero_ranges": (_int, [_p, _u32, _p, _p]),
    "nvo_bwd_zero_ranges": (_int, [_p, _p, _p, _p, _u32]),
    "nvo_fold_replicas": (_int, [_p, _u32, _p, _p, _p, _p]),
    "nvo_fused_adam_range": (_int, [_p, _p, _p]),
    "nvo_set_fused_adam": (_int, [_p, _p]),
    "nvo_ema_update": (_int, [_p, _u64, _p, _p, _p, _f, _u32, _p]),
    "nvo_ema_update_dev": (_int, [_p, _u64, _p, _p, _p, _f, _p, _p]),
    "nvo_ema_update_dev_part": (_int, [_p, _u64, _p, _p, _p, _f, _p, _p]),
    "nvo_cast_working_copy": (_int, [_p, _u64, _p, _p, _u32, _p, _p]),
    "nvo_adam_step_groups_mixed": (_int, [_p, _u32, _p, _p, _p, _p, _int, _p, _p, _f, _f, _f, _f, _f, _p, _u32, _p, _p]),
    "nvo_adam_step_groups_scaled": (_int, [_p, _u32, _p, _p, _p, _p, _int, _p, _p, _f, _f, _f, _f, _f, _p, _u32, _p, _p, _p]),
    "nvo_adam_step_groups_tail": (_int, [_p, _u32, _p, _p, _p, _p, _int, _p, _p, _f, _f, _f, _f, _f, _p, _u32, _p, _p, _p, _p]),
    "nvo_opt_commit_write": (_int, [_p, _u32, _u32, _u32, _p, _p, _p, _p, _f, _f, _u32, _f, _f, _p, _f, _f, _p, _u32, _p]),
    "nvo_opt_commit_table": (_int, [_p, _u32, _u32, _u32, _p, _p, _p, _p, _f, _f, _u32, _f, _f, _p, _f, _f, _p, _p, _u32, _p]),
    "nvo_opt_commit": (_int, [_p, _u32, _u32, _u32, _p, _p, _p, _p, _f, _f, _u32, _f, _f, _p, _f, _f]),
    "nvo_cast_bf16": (_int, [_p, _u64, _p, _p]),
    "nvo_cast_shards": (_int, [_p, _u64, _u32, _u32, _p, _p, _int, _p]),
    "nvo_flag_from_wire": (_int, [_p, _p, _p]),
}


def exported_symbols() -> list[str]:
    return sorted(_SIGNATURES)


def lib() -> C.CDLL:
    """Load (once) and return the native library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `python nerf-vo_amd/build.py`). "
                "There is no CPU fallback."
            )
        # PyTorch bundles its own libamdhip64.so.7 (+ HSA runtime); this library is linked against the
        # SONAME only.  Whichever copy is loaded first serves the whole process, and mixing the system
        # HIP runtime with torch's bundled HSA runtime ends in "no ROCm-capable device is detected".
        # Import torch FIRST so that both sides share torch's runtime (streams and pointers are then
        # handles of one and the same runtime instance).
        import torch  # noqa: F401

        handle = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().nvo_last_error()
        raise RuntimeError(f"nerfvo_hip {what} failed (code {rc}): {msg.decode() if msg else '?'}")
