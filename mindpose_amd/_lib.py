"""ctypes binding of ``libmindpose_hip.so`` (C ABI declared in ``include/mindpose_hip.h``).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
PyTorch is used only for device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MINDPOSE_HIP_LIB", os.path.join(_HERE, "csrc", "libmindpose_hip.so"))

MP_REFINE_NONE, MP_REFINE_SHIFT, MP_REFINE_DARK = 0, 1, 2
MP_CONV_SHARES_CUS = 1  # mp_conv_desc.flags: the launch runs beside other kernels of a training step (include/mindpose_hip.h)
MP_CONV_PHASES4 = 2     # mp_conv_desc.flags (fp16 family): the four 2x2 phase convs of a stride-2 3x3 data gradient as one launch

c_f32p = ctypes.c_void_p  # device pointers travel as integers
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t


class ConvDesc(ctypes.Structure):
    """``mp_conv_desc`` of include/mindpose_hip.h."""
    _fields_ = [(name, ctypes.c_int32) for name in (
        "n", "cin", "h", "w", "cout", "kh", "kw", "stride", "pad_top", "pad_left", "conv_h", "conv_w",
        "out_h", "out_w", "out_mul", "out_rep", "out_off_y", "out_off_x", "relu", "flags")]


class ConvStats(ctypes.Structure):
    """``mp_f16_conv_stats`` of include/mindpose_hip.h (BatchNorm partial sums from a conv launch's epilogue)."""
    _fields_ = [("mode", ctypes.c_int), ("relu", ctypes.c_int), ("partials", ctypes.c_void_p), ("partials_bytes", ctypes.c_size_t),
                ("z", ctypes.c_void_p), ("y", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("invstd", ctypes.c_void_p),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("pre_scale", ctypes.c_void_p), ("pre_shift", ctypes.c_void_p),
                ("pre_out", ctypes.c_void_p), ("pre_relu", ctypes.c_int)]


class MindposeHipError(RuntimeError):
    pass


class BnFwdJob(ctypes.Structure):  # mp_f16_bn_fwd_job (include/mindpose_hip.h)
    _fields_ = [(k, ctypes.c_void_p) for k in ("z", "gamma", "beta", "res", "y", "save_mean", "save_invstd", "moving_mean", "moving_var",
                                                "partials", "workspace")] + [("workspace_bytes", ctypes.c_size_t)] + \
               [(k, ctypes.c_int) for k in ("n", "c", "hw", "relu", "n_parts", "reserved")]


class BnBwdJob(ctypes.Structure):  # mp_f16_bn_bwd_job
    _fields_ = [(k, ctypes.c_void_p) for k in ("g", "z", "gamma", "save_mean", "save_invstd", "dz", "dgamma", "dbeta", "dgamma_acc",
                                                "dbeta_acc", "partials", "workspace")] + [("workspace_bytes", ctypes.c_size_t)] + \
               [(k, ctypes.c_int) for k in ("n", "c", "hw", "n_parts")]


_PROTOTYPES = {
    "mp_version": (ctypes.c_char_p, []),
    "mp_error_string": (ctypes.c_char_p, [c_int]),
    "mp_last_hip_error": (c_int, []),
    "mp_decode_topdown": (c_int, [c_f32p] * 7 + [c_int] * 7 + [ctypes.c_float, c_f32p, c_int, ctypes.c_void_p]),
    "mp_decode_topdown_debug": (c_int, [c_f32p] * 7 + [c_int] * 7 + [ctypes.c_float, c_f32p, c_int, c_f32p, ctypes.c_void_p]),
    "mp_flip_aggregate": (c_int, [c_f32p] * 4 + [c_int] * 5 + [ctypes.c_void_p]),
    "mp_flip_aggregate_decode": (c_int, [c_f32p] * 3 + [c_int] + [c_f32p] * 7 + [c_int] * 7
                                 + [ctypes.c_float, c_f32p, c_int, ctypes.c_void_p]),
    "mp_gaussian_target": (c_int, [c_f32p, c_f32p, c_int, c_f32p, c_f32p, c_f32p] + [c_int] * 4
                           + [ctypes.c_double] * 3 + [c_int, ctypes.c_void_p]),
    "mp_joints_mse_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mp_joints_mse_fwd": (c_int, [c_f32p] * 5 + [c_size_t] + [c_int] * 3 + [ctypes.c_void_p]),
    "mp_joints_mse_bwd": (c_int, [c_f32p] * 5 + [c_int] * 3 + [ctypes.c_void_p]),
    "mp_conv_packed_weight_bytes": (c_size_t, [c_int] * 4),
    "mp_conv_pack_weight": (c_int, [c_f32p, c_f32p] + [c_int] * 7 + [ctypes.c_void_p]),
    "mp_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc)] + [c_f32p] * 7 + [ctypes.c_void_p]),
    "mp_conv2d_fwd_variant": (c_int, [ctypes.POINTER(ConvDesc), c_int] + [c_f32p] * 7 + [ctypes.c_void_p]),
    "mp_conv_winograd_packed_weight_bytes": (c_size_t, [c_int] * 2),
    "mp_conv_winograd_pack_weight": (c_int, [c_f32p, c_f32p, c_int, c_int, ctypes.c_void_p]),
    "mp_conv_winograd_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "mp_conv2d_winograd_fwd": (c_int, [ctypes.POINTER(ConvDesc)] + [c_f32p] * 7 + [ctypes.c_void_p]),
    "mp_plan_add_conv_winograd": (c_int, [ctypes.c_void_p, ctypes.POINTER(ConvDesc)] + [c_f32p] * 7),
    "mp_plan_add_conv_variant": (c_int, [ctypes.c_void_p, ctypes.POINTER(ConvDesc), c_int] + [c_f32p] * 7),
    "mp_deconv4x4s2_gemm_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "mp_deconv4x4s2_gemm_fwd": (c_int, [ctypes.POINTER(ConvDesc)] + [c_f32p] * 5 + [ctypes.c_void_p]),
    "mp_plan_add_deconv4x4s2_gemm": (c_int, [ctypes.c_void_p, ctypes.POINTER(ConvDesc)] + [c_f32p] * 5),
    "mp_maxpool3x3s2_same": (c_int, [c_f32p, c_f32p] + [c_int] * 4 + [ctypes.c_void_p]),
    "mp_fuse_upsample_sum": (c_int, [c_f32p, c_f32p, c_int, c_f32p, c_int, c_f32p, c_int, c_f32p] + [c_int] * 5 + [ctypes.c_void_p]),
    "mp_plan_add_fuse_sum": (c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_int, c_f32p, c_int, c_f32p, c_int, c_f32p] + [c_int] * 5),
    "mp_plan_create": (ctypes.c_void_p, []),
    "mp_plan_destroy": (None, [ctypes.c_void_p]),
    "mp_plan_add_conv": (c_int, [ctypes.c_void_p, ctypes.POINTER(ConvDesc)] + [c_f32p] * 7),
    "mp_plan_add_maxpool": (c_int, [ctypes.c_void_p, c_f32p, c_f32p] + [c_int] * 4),
    "mp_plan_size": (c_int, [ctypes.c_void_p]),
    "mp_plan_run": (c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mp_plan_run_range": (c_int, [ctypes.c_void_p, c_int, c_int, ctypes.c_void_p]),
    "mp_plan_entry_info": (c_int, [ctypes.c_void_p, c_int, ctypes.POINTER(ctypes.c_int64)]),
    "mp_debug_set_stamp_buffer": (c_int, [ctypes.c_void_p, c_size_t]),
    "mp_bn_workspace_bytes": (c_size_t, [c_int]),
    "mp_bn_train_fwd": (c_int, [c_f32p] * 9 + [c_int] * 3 + [ctypes.c_float, ctypes.c_float, c_int, c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_bn_train_bwd": (c_int, [c_f32p] * 10 + [c_int] * 4 + [c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_bn_train_bwd_acc": (c_int, [c_f32p] * 12 + [c_int] * 4 + [c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_fuse_upsample_sum_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_int, c_f32p, c_int] + [c_int] * 5 + [ctypes.c_void_p]),
    "mp_conv_wgrad_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "mp_conv_wgrad": (c_int, [ctypes.POINTER(ConvDesc), c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_adamw_step": (c_int, [c_f32p] * 4 + [c_size_t] + [ctypes.c_float] * 5 + [ctypes.c_void_p]),
    "mp_adamw_step_scaled": (c_int, [c_f32p] * 4 + [c_size_t] + [ctypes.c_float] * 6 + [c_f32p, ctypes.c_void_p]),
    "mp_grad_finite_check": (c_int, [c_f32p, c_size_t, c_f32p, ctypes.c_void_p]),
    "mp_comm_available": (c_int, []),
    "mp_comm_last_error": (c_int, []),
    "mp_comm_get_unique_id": (c_int, [ctypes.c_void_p]),
    "mp_comm_init_rank": (c_int, [ctypes.POINTER(ctypes.c_void_p), c_int, ctypes.c_void_p, c_int]),
    "mp_comm_destroy": (c_int, [ctypes.c_void_p]),
    "mp_comm_count": (c_int, [ctypes.c_void_p]),
    "mp_allreduce_grads": (c_int, [ctypes.c_void_p, c_f32p, c_size_t, c_int, ctypes.c_void_p]),
    "mp_reduce_scatter_allgather_grads": (c_int, [ctypes.c_void_p, c_f32p, c_size_t, c_int, c_int, c_int, ctypes.c_void_p]),
    "mp_f16_bn_train_fwd": (c_int, [c_f32p] * 9 + [c_int] * 3 + [ctypes.c_float, ctypes.c_float, c_int, c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_sum_tensors": (c_int, [c_f32p] * 5 + [c_size_t, c_int, ctypes.c_void_p]),
    "mp_f16_bn_train_bwd": (c_int, [c_f32p] * 13 + [c_int] * 4 + [c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_f16_fuse_upsample_sum_bwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_int, c_f32p, c_int, c_f32p, c_int] + [c_int] * 5 + [ctypes.c_void_p]),
    "mp_f16_conv_wgrad_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "mp_f16_conv_wgrad": (c_int, [ctypes.POINTER(ConvDesc), c_f32p, c_f32p, c_f32p, ctypes.c_float, c_int, c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_f16_conv_wgrad_grouped_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    "mp_f16_conv_wgrad_grouped": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                          ctypes.POINTER(ctypes.c_void_p), c_int, ctypes.c_float, c_int, c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_maxpool3x3s2_same_bwd": (c_int, [c_f32p] * 3 + [c_int] * 4 + [ctypes.c_void_p]),
    "mp_stem_conv_wgrad": (c_int, [c_f32p] * 3 + [c_int] * 6 + [ctypes.c_void_p]),
    "mp_gather_phase": (c_int, [c_f32p] * 2 + [c_int] * 6 + [ctypes.c_void_p]),
    "mp_f16_gather_phase": (c_int, [c_f32p] * 2 + [c_int] * 6 + [ctypes.c_void_p]),
    "mp_plan_set_lane": (c_int, [ctypes.c_void_p, c_int]),
    "mp_plan_add_barrier": (c_int, [ctypes.c_void_p]),
    "mp_flip_width": (c_int, [c_f32p, c_f32p] + [c_int] * 4 + [ctypes.c_void_p]),
    "mp_warp_affine": (c_int, [c_f32p] * 6 + [c_int] * 4 + [ctypes.POINTER(ctypes.c_float)] * 2 + [ctypes.c_void_p]),
    # fp16 matrix-core inference path (channel-blocked activations)
    "mp_f16_packed_weight_bytes": (c_size_t, [c_int] * 4),
    "mp_f16_activation_bytes": (c_size_t, [c_int] * 4),
    "mp_f16_pack_weight": (c_int, [c_f32p, c_f32p] + [c_int] * 7 + [ctypes.c_void_p]),
    "mp_f16_pack_weight_batch": (c_int, [c_f32p, c_f32p, c_int, ctypes.c_uint, ctypes.c_void_p]),
    "mp_conv_pack_weight_batch": (c_int, [c_f32p, c_f32p, c_int, ctypes.c_uint, ctypes.c_void_p]),
    "mp_f16_to_c8": (c_int, [c_f32p, c_f32p] + [c_int] * 4 + [ctypes.c_void_p]),
    "mp_f16_from_c8": (c_int, [c_f32p, c_f32p] + [c_int] * 4 + [ctypes.c_void_p]),
    "mp_f16_conv2d_fwd": (c_int, [ctypes.POINTER(ConvDesc), c_int] + [c_f32p] * 7 + [ctypes.c_void_p]),
    "mp_f16_conv_stats_parts": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "mp_f16_conv2d_fwd_stats": (c_int, [ctypes.POINTER(ConvDesc), c_int] + [c_f32p] * 6 + [ctypes.POINTER(ConvStats), ctypes.c_void_p]),
    "mp_f16_bn_train_fwd_stats": (c_int, [c_f32p] * 9 + [c_int] * 3 + [ctypes.c_float, ctypes.c_float, c_int, c_f32p, c_int, c_f32p,
                                          c_size_t, ctypes.c_void_p]),
    "mp_f16_conv_pre_supported": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "mp_f16_conv_supported": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_int, c_int]),
    "mp_f16_bn_train_finalize": (c_int, [c_f32p] * 6 + [c_int] * 3 + [ctypes.c_float, ctypes.c_float, c_f32p, c_int, c_f32p, c_f32p, c_f32p,
                                         c_size_t, ctypes.c_void_p]),
    "mp_f16_bn_train_bwd_stats": (c_int, [c_f32p] * 10 + [c_int] * 3 + [c_f32p, c_int, c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_f16_bn_train_fwd_stats_grouped": (c_int, [ctypes.POINTER(BnFwdJob), c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mp_f16_bn_train_bwd_stats_grouped": (c_int, [ctypes.POINTER(BnBwdJob), c_int, ctypes.c_void_p]),
    "mp_f16_ew_stats_parts": (c_int, [c_int] * 3),
    "mp_f16_fuse_term_stats_parts": (c_int, [c_int] * 5),
    "mp_f16_sum_tensors_stats": (c_int, [c_f32p] * 7 + [c_int] * 4 + [c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_f16_fuse_sum_bwd_term_stats": (c_int, [c_f32p] * 3 + [c_int] * 6 + [c_f32p, c_f32p, c_int, c_f32p, c_size_t, ctypes.c_void_p]),
    "mp_optimizer_step": (c_int, [c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_size_t, ctypes.c_float,
                                  ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_float * 5), ctypes.c_void_p]),
    "mp_f16_basicblock_fwd": (c_int, [c_f32p] * 8 + [c_int] * 5 + [ctypes.c_void_p]),
    "mp_f16_basicblock_supported": (c_int, [c_int] * 4),
    "mp_stem_conv_fwd": (c_int, [c_f32p] * 4 + [c_int, c_f32p] + [c_int] * 3 + [ctypes.c_void_p]),
    "mp_plan_add_stem_conv": (c_int, [ctypes.c_void_p] + [c_f32p] * 4 + [c_int, c_f32p] + [c_int] * 3),
    "mp_f16_stem_conv_fwd": (c_int, [c_f32p] * 4 + [c_int, c_f32p] + [c_int] * 3 + [ctypes.c_void_p]),
    "mp_plan_add_stem_conv_f16": (c_int, [ctypes.c_void_p] + [c_f32p] * 4 + [c_int, c_f32p] + [c_int] * 3),
    "mp_f16_dual_pw_fwd": (c_int, [c_f32p] * 4 + [c_int] + [c_f32p] * 3 + [c_int] + [c_f32p] * 2 + [c_int] * 6 + [ctypes.c_void_p]),
    "mp_plan_add_dual_pw_f16": (c_int, [ctypes.c_void_p] + [c_f32p] * 4 + [c_int] + [c_f32p] * 3 + [c_int] + [c_f32p] * 2 + [c_int] * 6),
    "mp_f16_expand_reduce_fwd": (c_int, [c_f32p] * 5 + [c_int] + [c_f32p] * 3 + [c_int] + [c_f32p] * 2 + [c_int] * 6 + [ctypes.c_void_p]),
    "mp_plan_add_expand_reduce_f16": (c_int, [ctypes.c_void_p] + [c_f32p] * 5 + [c_int] + [c_f32p] * 3 + [c_int] + [c_f32p] * 2 + [c_int] * 6),
    "mp_f16_ds_expand_reduce_fwd": (c_int, [c_f32p] * 8 + [c_int] + [c_f32p] * 3 + [c_int] + [c_f32p] * 2 + [c_int] * 6 + [ctypes.c_void_p]),
    "mp_plan_add_ds_expand_reduce_f16": (c_int, [ctypes.c_void_p] + [c_f32p] * 8 + [c_int] + [c_f32p] * 3 + [c_int] + [c_f32p] * 2 + [c_int] * 6),
    "mp_expand_reduce_fwd": (c_int, [c_f32p] * 14 + [c_int] * 6 + [ctypes.c_void_p]),
    "mp_plan_add_expand_reduce": (c_int, [ctypes.c_void_p] + [c_f32p] * 14 + [c_int] * 6),
    "mp_plan_add_basicblock_f16": (c_int, [ctypes.c_void_p] + [c_f32p] * 8 + [c_int] * 5),
    "mp_f16_fuse_upsample_sum": (c_int, [c_f32p, c_f32p, c_int, c_f32p, c_int, c_f32p, c_int, c_f32p] + [c_int] * 5 + [ctypes.c_void_p]),
    "mp_plan_add_conv_f16": (c_int, [ctypes.c_void_p, ctypes.POINTER(ConvDesc), c_int] + [c_f32p] * 7),
    "mp_plan_add_fuse_sum_f16": (c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_int, c_f32p, c_int, c_f32p, c_int, c_f32p] + [c_int] * 5),
    "mp_plan_add_layout_f16": (c_int, [ctypes.c_void_p, c_int, c_f32p, c_f32p] + [c_int] * 4),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_lib = None


def load():
    """Load the shared library (once).  Raises ``MindposeHipError`` if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MindposeHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C mindpose_amd/csrc -j8`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        lib = load()
        msg = lib.mp_error_string(rc).decode()
        extra = f" (hipError {lib.mp_last_hip_error()})" if rc == -4 else ""
        raise MindposeHipError(f"{what} failed: {msg}{extra}")


def ptr(t):
    """Device pointer of a contiguous fp32/int32/fp64 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    t = getattr(t, "c8_tensor", t)  # channel-blocked fp16 activation wrapper
    if not t.is_cuda:
        raise MindposeHipError("tensor must live on the GPU: the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise MindposeHipError("tensor must be contiguous")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda_f32(t, name):
    if not torch.is_tensor(t):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise MindposeHipError(f"{name} must be a CUDA tensor: the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
