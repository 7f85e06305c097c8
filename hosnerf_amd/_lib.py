"""ctypes binding of libhosrender.so (the C ABI declared in include/hosrender.h).

There is deliberately no fallback: if the shared library has not been built, or a tensor handed
to an op is not a contiguous fp32 HIP tensor, this module raises.  The product path never
routes through oracle/ or any CPU implementation.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HOS_LIB_PATH", os.path.join(_HERE, "lib", "libhosrender.so"))   # override: timing experiments only


class HosLibraryError(RuntimeError):
    pass


_P, _I, _F, _L = c_void_p, c_int, c_float, c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/hosrender.h one to one
PROTOTYPES = {
    "hos_version": [],
    "hos_device_count": [],
    "hos_error_string": [_I],
    "hos_set_gemm_mode": [_I],
    "hos_set_thread_gemm_mode": [_I],
    "hos_get_gemm_mode": [],
    "hos_set_range_flag": [_P],
    "hos_linear_fwd": [_P, _I, _I, _P, _I, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _F, _F, _P, _P],
    "hos_linear_fwd_splitk": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "hos_linear_fwd_splitk_ws_floats": [_I, _I, _I],
    "hos_linear_fwd_splitk_det": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _L, _P],
    "hos_linear_dgrad": [_P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "hos_linear_wgrad": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "hos_thin_linear_fwd": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P],
    "hos_thin_linear_dgrad": [_P, _I, _P, _I, _I, _P, _I, _P, _P, _I, _I, _I, _P],
    "hos_canonical_fold_pack": [_P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "hos_canonical_fold_unfold": [_P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "hos_mlp_bwd_defer": [_I],
    "hos_mlp_bwd_flush": [_P],
    "hos_mlp_bwd_ws_floats": [_I, _I, _I, _I],
    "hos_linear_wgrad_tr": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _L, _P],
    "hos_linear_bwd_fused": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _L, _P, _P],
    "hos_stage1_loss_fwd": [_P, _P, _I, _P, _P, _I, _P, _F, _F, _F, _F, _P, _P],
    "hos_stage1_loss_bwd": [_P, _P, _I, _I, _P, _P, _F, _F, _F, _F, _P, _P, _P, _P],
    "hos_lpips_prep": [_P, _L, _P, _P],
    "hos_im2col3x3": [_P, _I, _I, _I, _I, _P, _I, _P],
    "hos_col2im3x3": [_P, _I, _I, _I, _I, _I, _P, _P, _P],
    "hos_maxpool2x2_fwd": [_P, _I, _I, _I, _I, _P, _P],
    "hos_maxpool2x2_bwd": [_P, _P, _I, _I, _I, _I, _P, _P],
    "hos_lpips_head_fwd": [_P, _P, _I, _I, _I, _F, _P, _P],
    "hos_lpips_head_bwd": [_P, _P, _I, _I, _I, _F, _P, _I, _P, _P],
    "hos_lpips_finish": [_P, _I, _P, _P],
    "hos_lpips_part_floats": [_I],
    "hos_bias_relu": [_P, _P, _L, _I, _P],
    "hos_unpack_patches_fwd": [_P, _P, _P, _F, _L, _P, _P],
    "hos_unpack_patches_bwd": [_P, _P, _L, _F, _F, _F, _P, _P],
    "hos_camera_rays": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
    "hos_rays_aabb": [_P, _P, _L, _P, _P, _P, _P, _P],
    "hos_deconv3d_col2im": [_P, _P, _I, _I, _F, _I, _P, _P],
    "hos_deconv3d_im2col": [_P, _I, _I, _P, _P],
    "hos_bias_lrelu": [_P, _P, _L, _I, _F, _I, _P],
    "hos_shard_interleave": [_P, _I, _I, _I, _P, _P],
    "hos_deconv3d_dpre": [_P, _P, _L, _I, _F, _I, _P, _P, _P],
    "hos_gemv_ws_floats": [_I, _I],
    "hos_gemv_rowvec": [_P, _P, _I, _I, _I, _P, _I, _F, _I, _P, _P, _P],
    "hos_outer_accum": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P],
    "hos_split_planes": [_P, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    "hos_linearp_fwd": [_P, _I, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _I, _P, _I, _F, _P],
    "hos_split_planes_t_batch": [_I, _P, _P, _P, _P, _P, _P, _P],
    "hos_split_planes2": [_P, _I, _I, _I, _P, _I, _P, _I, _P],
    "hos_planes_rowdot": [_P, _I, _I, _P, _P, _F, _I, _L, _P, _P],
    "hos_planes_rowdot_b": [_P, _I, _I, _P, _P, _F, _I, _L, _P, _P],
    "hos_linearp_fwd_b": [_P, _I, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _P, _I, _F, _P],
    "hos_linearp_dgrad": [_P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P, _I, _P],
    "hos_linearp_wgrad": [_P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _P, _L, _P],
    "hos_resample": [_P, _P, _I, _I, _I, _F, _F, _P, _F, _F, _P, _P, _F, _F, _F, _P, _P, _P, _P],
    "hos_encode_ipe": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P],
    "hos_encode_ipe_planes": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P],
    "hos_encode_viewdirs": [_P, _I, _I, _P, _I, _I, _P],
    "hos_alpha_weights_fwd": [_P, _P, _P, _I, _I, _I, _P, _P],
    "hos_alpha_weights_bwd": [_P, _P, _P, _P, _I, _I, _I, _P, _P],
    "hos_volrender_fwd": [_P, _P, _I, _I, _F, _P, _P],
    "hos_volrender_bwd": [_P, _P, _P, _I, _I, _F, _P, _P, _P],
    "hos_interlevel_fwd": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "hos_interlevel_bwd": [_P, _P, _P, _P, _I, _I, _I, _F, _P, _P],
    "hos_distortion_fwd": [_P, _P, _I, _I, _P, _P],
    "hos_distortion_bwd": [_P, _P, _I, _I, _F, _P, _P],
    "hos_head_grad": [_P, _P, _P, _P, _I, _F, _P, _I, _I, _P, _I, _P],
    "hos_human_sample_warp": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    "hos_lbs_forward": [_P, _P, _P, _P, _I, _I, _P, _P, _L, _I, _P, _P, _P],
    "hos_embed_hannw": [_P, _P, _I, _P, _I, _L, _P, _I, _P, _I, _P, _P],
    "hos_embed_fourier": [_P, _I, _P, _I, _L, _P, _I, _P, _I, _P],
    "hos_human_sample_warp_bwd": [_P, _P, _P, _P, _I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "hos_lbs_forward_bwd": [_P, _P, _P, _P, _I, _I, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P, _P],
    "hos_embed_bwd": [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I, _L, _P, _I, _P, _P],
    "hos_slice_mask": [_P, _I, _I, _P, _I, _I, _L, _I, _P, _I, _P, _P],
    "hos_slice_pad": [_P, _I, _I, _L, _I, _P, _I, _P, _P],
    "hos_rgbsigma_grad": [_P, _P, _L, _P, _I, _P],
    "hos_raw2outputs_fwd": [_P, _I, _P, _I, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P],
    "hos_raw2outputs_bwd": [_P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _F, _I, _I, _P, _I, _P, _I, _P, _P],
    "hos_merge_composite_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P],
    "hos_merge_composite_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "hos_mlp_chain_weight_bytes": [],
    "hos_mlp_chain_aux_floats": [],
    "hos_mlp_chain_pack": [_P, _P, _P, _P, _P, _P],
    "hos_mlp_chain_pack_fold": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "hos_mlp_chain_unfold_grad": [_P, _P, _P, _I, _I, _P, _I, _P, _P],
    "hos_mlp_chain128_fwd": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _L, _P, _P],
    "hos_mlp_chain_bwd_steps": [_I],
    "hos_mlp_chain_bwd_image_bytes": [_I, _I],
    "hos_mlp_chain_bwd_ws_floats": [_I, _I],
    "hos_mlp_chain_bwd_pack": [_I, _P, _P, _P, _P, _P, _P, _P, _P],
    "hos_mlp_chain_bwd": [_I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P],
    "hos_compact_workspace_ints": [],
    "hos_compact_rows": [_P, _F, _P, _P, _L, _P, _P, _P, _P, _P, _P],
    "hos_scatter_rows": [_P, _P, _P, _L, _P, _P],
    "hos_pose_refine_saved_floats": [],
    "hos_pose_refine_fwd": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    "hos_pose_refine_workspace_floats": [],
    "hos_pose_refine_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "hos_motion_basis_fwd": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
    "hos_motion_basis_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "hos_train_losses_workspace_floats": [],
    "hos_train_losses_fwd": [_P, _P, _L, _F, _F, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _P, _F, _F, _F, _P, _P, _P],
    "hos_train_losses_bwd": [_P, _P, _P, _P, _L, _F, _P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _P, _F, _F, _F, _P, _P, _P, _P, _P],
    "hos_sumsq": [_P, _L, _P, _P],
    "hos_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P, _F, _P],
    "hos_adam_step_dyn": [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P, _F, _P],
    "hos_rowdot_lrelu_fwd": [_P, _P, _I, _P, _I, _I, _F, _P, _P],
    "hos_rowdot_lrelu_bwd": [_P, _P, _P, _P, _I, _I, _I, _F, _P, _I, _P, _P, _P],
    "hos_volume_softmax_fwd": [_P, _P, _I, _L, _P, _P],
    "hos_volume_softmax_bwd": [_P, _P, _I, _L, _P, _P],
    "hos_volume_channel_last": [_P, _I, _L, _P, _P],
    "hos_volume_pair_bwd": [_P, _P, _I, _I, _L, _P, _P],
    "hos_copy_or_zero_n": [_I, _P, _P, _P, _P],
    "hos_debug_stamp": [_P, _I, _P],
    "hos_clear_last_error": [],
    "hos_add_n": [_I, _P, _L, _P, _P],
    "hos_any_abs_below": [_P, _L, _F, _P, _P],
    "hos_state_embed_grad": [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    "hos_embed_bwd_res": [_P, _P, _I, _I, _P, _I, _I, _P, _I, _I, _L, _P, _P, _P, _P],
    "hos_head_grad_padded": [_P, _P, _P, _P, _I, _F, _P, _I, _I, _P, _I, _P],
    "hos_sumsq_blocks": [],
    "hos_sumsq_partials": [_I, _P, _P, _P, _P],
    "hos_adam_multi": [_I, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _P, _F, _P, _P, _P],
    "hos_adam_multi_lazy": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _P, _F, _P, _P, _P],
    "hos_adam_lazy_prepare": [_I, _P, _P, _P, _F, _F, _P, _P],
}
_RESTYPES = {"hos_error_string": c_char_p, "hos_mlp_bwd_ws_floats": c_int64, "hos_train_losses_workspace_floats": c_int64,
             "hos_pose_refine_saved_floats": c_int64, "hos_compact_workspace_ints": c_int64,
             "hos_pose_refine_workspace_floats": c_int64, "hos_mlp_chain_weight_bytes": c_int64,
             "hos_mlp_chain_aux_floats": c_int64, "hos_gemv_ws_floats": c_int64,
             "hos_mlp_chain_bwd_image_bytes": c_int64, "hos_mlp_chain_bwd_ws_floats": c_int64,
             "hos_linear_fwd_splitk_ws_floats": c_int64}

_lib = None


def load() -> ctypes.CDLL:
    """Load the library once; raise HosLibraryError with a build hint if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HosLibraryError(
            f"{LIB_PATH} not found: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "hosnerf_amd has no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HosLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HosLibraryError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().hos_error_string(int(code))
        raise HosLibraryError(f"{what} failed with code {code}: {msg.decode() if msg else '?'}")


def require_gpu():
    if not torch.cuda.is_available():
        raise HosLibraryError("no HIP device visible: hosnerf_amd ops run on MI355X only (no CPU fallback)")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t, dtype=torch.float32) -> int:
    """Device pointer of a contiguous HIP tensor of the expected dtype (None -> NULL)."""
    if t is None:
        return 0
    if not isinstance(t, torch.Tensor):
        raise HosLibraryError(f"expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise HosLibraryError("hosnerf_amd ops take HIP (device='cuda') tensors only -- there is no CPU path")
    if t.dtype != dtype:
        raise HosLibraryError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise HosLibraryError("expected a contiguous tensor")
    return t.data_ptr()


def call(name: str, *args):
    """Invoke an entry point on the current stream (appended as the last argument) and check it."""
    lib = load()
    code = getattr(lib, name)(*args, stream_ptr())
    check(code, name)
