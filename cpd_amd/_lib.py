"""ctypes binding of cpd_amd/csrc/libcpd_hip.so (the C-ABI of include/cpd_hip.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the ops raises.
torch is imported first on purpose -- it brings the process's HIP runtime (libamdhip64.so.7), which
libcpd_hip.so then binds to, so both share streams and device memory.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPD_HIP_LIB", os.path.join(_HERE, "csrc", "libcpd_hip.so"))  # override: diagnostics only

CPD_ERRORS = {-1: "CPD_ERR_ARG", -2: "CPD_ERR_WORKSPACE", -3: "CPD_ERR_LAUNCH", -4: "CPD_ERR_UNSUPPORTED"}


class CpdHipError(RuntimeError):
    pass


_lib = None

_VP, _I, _F, _SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_D = ctypes.c_double
_I3 = ctypes.POINTER(ctypes.c_int32)
_FP = ctypes.POINTER(ctypes.c_float)

# name -> (restype, argtypes): one row per declaration in include/cpd_hip.h
SIGNATURES = {
    "cpd_launch_log_enable": (None, [_I]),
    "cpd_launch_log_note": (None, [ctypes.c_char_p]),
    "cpd_launch_log_dump": (_SZ, [ctypes.c_char_p, _SZ]),
    "cpd_version": (ctypes.c_char_p, []),
    "cpd_last_hip_error": (_I, []),
    "cpd_voxel_grid_size": (_I, [_FP, _FP, _I3]),
    "cpd_voxelize_workspace_bytes": (_SZ, [_I, _I, _I, _FP, _FP]),
    "cpd_voxelize_batch_workspace_bytes": (_SZ, [_I, _I, _I, _I, _FP, _FP]),
    "cpd_voxelize_batch": (_I, [_VP, _I3, _I, _I, _FP, _FP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_voxelize_batch_index": (_I, [_VP, _I3, _I, _I, _FP, _FP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP, _SZ, _I, _VP]),
    "cpd_voxelize_batch_canonical": (_I, [_VP, _I3, _I, _I, _FP, _FP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP, _SZ, _I, _VP]),
    "cpd_voxelize_batch_frames": (_I, [_VP, _I3, _I, _I, _FP, _FP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP, _SZ, _I, _I, _VP]),
    "cpd_voxelize": (_I, [_VP, _I, _I, _FP, _FP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_index_bytes": (_SZ, [_I, _I3, _I]),
    "cpd_index_build": (_I, [_VP, _I, _I, _I3, _VP, _SZ, _VP]),
    "cpd_order_rows_by_taps": (_I, [_VP, _I, _I, _I3, _I3, _VP, _I, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_index_set_order": (_I, [_VP, _VP, _VP]),
    "cpd_order_rows_bricks": (_I, [_VP, _I, _I, _I3, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_rulebook_chunk_ordered": (_I, [_VP, _VP, _I, _I, _I3, _I3, _I3, _I3, _VP, _I, _VP, _VP, _VP]),
    "cpd_rulebook_plan_bytes": (_SZ, [_I, _I, _I]),
    "cpd_rulebook_plan": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "cpd_gather_conv_planned_supported": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "cpd_gather_conv_planned": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _I, _VP, _VP]),
    "cpd_rulebook_subm": (_I, [_VP, _I, _I, _I3, _I3, _VP, _VP, _VP, _VP]),
    "cpd_conv_out_shape": (_I, [_I3, _I3, _I3, _I3, _I3]),
    "cpd_conv_outset": (_I, [_VP, _I, _I, _I3, _I3, _I3, _I3, _VP, _SZ, _VP, _VP]),
    "cpd_index_emit": (_I, [_VP, _I, _I3, _VP, _I, _VP]),
    "cpd_rulebook_conv": (_I, [_VP, _I, _I, _I3, _I3, _I3, _I3, _VP, _VP, _VP, _VP]),
    "cpd_packed_weight_floats": (_SZ, [_I, _I, _I]),
    "cpd_pack_weight": (_I, [_VP, _I, _I, _I, _VP, _VP]),
    "cpd_gather_conv": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _VP, _I, _I, _VP]),
    "cpd_gather_conv_scaled": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _VP, _I, _I, _VP, _VP]),
    "cpd_gather_conv_tile": (_I, [_I, _I, _I, _I, _I] + [ctypes.POINTER(_I)] * 4),
    "cpd_conv3x3_rows_tile": (_I, [_I, _I, _I, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "cpd_conv3x3_rows_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "cpd_conv3x3_rows": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _I, _VP]),
    "cpd_conv3x3_rows_scaled": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _I, _VP, _VP]),
    "cpd_conv3x3_rows_ranged": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _I, _VP, _VP, _VP]),
    "cpd_gather_conv_ranged": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _VP, _I, _I, _VP, _VP, _VP]),
    "cpd_gather_conv_ws": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _VP, _I, _I, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_gather_conv_split_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "cpd_absmax_rows": (_I, [_VP, _I, ctypes.c_longlong, _I, _VP, _VP]),
    "cpd_densify_nchw": (_I, [_VP, _VP, _I, _I, _I, _I3, _VP, _VP]),
    "cpd_densify_nhwc": (_I, [_VP, _VP, _I, _I, _I, _I3, _VP, _VP]),
    "cpd_densify_nhwc_cd": (_I, [_VP, _VP, _I, _I, _I, _I3, _VP, _VP]),
    "cpd_densify_nhwc_rows": (_I, [_VP, _VP, _I, _I, _I, _I3, _VP, _VP]),
    "cpd_densify_nhwc_clear": (_I, [_VP, _I, _I, _I, _I3, _VP, _VP]),
    "cpd_rulebook_conv2d": (_I, [_I, _I, _I, _I, _I, _I, _I, _VP, _VP]),
    "cpd_center_decode_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "cpd_center_decode": (_I, [_VP, _VP, _VP, _VP, _VP, _I, ctypes.c_longlong, _I, _I, _I, _I, _I, _I, _F, _FP, _FP, _FP, _F,
                               _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_boxes_overlap_bev": (_I, [_VP, _I, _VP, _I, _VP, _VP]),
    "cpd_boxes_iou_bev": (_I, [_VP, _I, _VP, _I, _VP, _VP]),
    "cpd_boxes_iou3d": (_I, [_VP, _I, _VP, _I, _VP, _VP]),
    "cpd_nms_workspace_bytes": (_SZ, [_I]),
    "cpd_nms_rotated": (_I, [_VP, _I, _F, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_nms_normal": (_I, [_VP, _I, _F, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_nms_batch": (_I, [_VP, _VP, _I, _I, _F, _I, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_nms_batch_first": (_I, [_VP, _VP, _I, _I, _F, _I, _I, _I, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_nms_batch_where": (_I, [_VP, _VP, _VP, _I, _I, _F, _I, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_rank_scores": (_I, [_VP, _I, _VP, _VP, _I, _I, _F, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "cpd_select_boxes": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "cpd_boxes_iou_bev_cpu": (_I, [_VP, _I, _VP, _I, _VP]),
    "cpd_col_reduce_workspace_bytes": (_SZ, [_I, _I]),
    "cpd_col_sum": (_I, [_VP, _I, _I, _I, _VP, _VP, _SZ, _VP]),
    "cpd_bn_stats": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_mask_points_workspace_bytes": (_SZ, [_I]),
    "cpd_mask_points_by_range": (_I, [_VP, _I, _I, _FP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_merge_sweeps": (_I, [_VP, _I3, _I, _I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _VP, _VP]),
    "cpd_points_in_boxes": (_I, [_I, _I, _I, _VP, _VP, _I, _F, _VP, _VP]),
    "cpd_nearest_bev_iou": (_I, [_VP, _I, _VP, _I, _VP, _VP]),
    "cpd_anchor_assign_workspace_bytes": (_SZ, [_I, _I]),
    "cpd_anchor_assign": (_I, [_VP, _I, _VP, _I, _VP, _F, _F, _I, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_anchor_decode": (_I, [_VP, _VP, _VP, _I, _I, _I, _F, _F, _VP, _VP]),
    "cpd_points_in_boxes_mask": (_I, [_VP, _I, _VP, _I, _I, _VP, _VP]),
    "cpd_crop_boxes_workspace_bytes": (_SZ, [_I]),
    "cpd_crop_boxes": (_I, [_VP, _I, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_transform_points": (_I, [_VP, _I, _I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _I, _VP, _VP]),
    "cpd_atss_workspace_bytes": (_SZ, [_I, _I]),
    "cpd_atss_assign": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_mfma_burn_flops": (_D, [_I, _I]),
    "cpd_mfma_burn": (_I, [_VP, _VP, _VP, _I, _I, _VP]),
    "cpd_voxel2pinds": (_I, [_VP, _I, _I, _I3, _VP, _VP]),
    "cpd_voxel_query": (_I, [_I, _I, _I, _I, _I, _F, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "cpd_voxel_query_index": (_I, [_I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "cpd_voxel_query_index_grid": (_I, [_I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "cpd_roi_grid_points": (_I, [_VP, _I, _I, _I, _I, _FP, _FP, _I, _I3, ctypes.POINTER(ctypes.c_void_p), _I, _VP, _VP]),
    "cpd_group_points": (_I, [_I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "cpd_group_points_grad": (_I, [_I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "cpd_voxel_pool_max": (_I, [_I, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "cpd_voxel_pool_max_mlp": (_I, [_I, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _I, _VP]),
    "cpd_voxel_pool_max_mlp_ranged": (_I, [_I, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _I, _VP, _VP]),
    "cpd_bn_stats_finalize": (_I, [_VP, _I, _I, _I, _F, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_bn_finalize": (_I, [_VP, _VP, _I, _I, _F, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "cpd_pack_weight_adjoint": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "cpd_pack_batch_table_bytes": (_SZ, [_I]),
    "cpd_pack_batch_prepare": (_I, [_VP, _I, _VP, _SZ, _VP]),
    "cpd_pack_batch_run": (_I, [_VP, _I, _VP, _I, _VP]),
    "cpd_affine_rows": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _I, _VP]),
    "cpd_bn_bwd_reduce": (_I, [_VP, _I, _VP, _I, _VP, _I, _VP, _VP, _I, _I, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_bn_bwd_apply": (_I, [_VP, _I, _VP, _I, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I, _VP, _VP]),
    "cpd_bn_finalize_sync": (_I, [_VP, _VP, _VP, _I, _F, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "cpd_bn_bwd_apply_sync": (_I, [_VP, _I, _VP, _I, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I, _VP, _VP]),
    "cpd_relu_bwd": (_I, [_VP, _I, _VP, _I, _I, _I, _VP, _I, _VP]),
    "cpd_conv_wgrad_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "cpd_conv_wgrad": (_I, [_VP, _I, _I, _VP, _I, _I, _VP, _I, _I, _VP, _I, _VP, _SZ, _VP]),
    "cpd_conv_wgrad_scaled": (_I, [_VP, _I, _I, _VP, _I, _I, _VP, _I, _I, _VP, _I, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_rulebook_conv_transpose": (_I, [_VP, _I, _I, _I3, _I3, _I3, _I3, _VP, _VP, _VP]),
    "cpd_rulebook_conv2d_transpose": (_I, [_I, _I, _I, _I, _I, _I, _I, _VP, _VP]),
    "cpd_center_targets_workspace_bytes": (_SZ, [_I, _I]),
    "cpd_center_targets": (_I, [_VP, _I, _I, _I, _I, _I, _I, _FP, _FP, _I, _D, _I, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_center_loss_workspace_bytes": (_SZ, [_I, _I, _I]),
    "cpd_center_loss": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _I, _FP, _F, _F, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_anchor_loss_workspace_bytes": (_SZ, [_I, _I]),
    "cpd_anchor_loss": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _F, _FP, _F, _F, _F, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "cpd_adam_step": (_I, [_VP, _VP, _VP, _VP, _SZ, _F, _F, _F, _F, _F, _I, _F, _VP, _VP]),
}


def lib():
    """Load (once) and return the C-ABI library. Raises loudly when it is absent: the product has
    no eager/PyTorch/CPU fallback for these ops."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CpdHipError(
                "libcpd_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C cpd_amd/csrc`); there is no fallback path")
        cdll = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)  # AttributeError = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = cdll
    return _lib


def check(rc, what):
    if rc != 0:
        extra = ""
        if rc == -3:
            extra = " (hipError %d)" % lib().cpd_last_hip_error()
        raise CpdHipError("%s failed: %s%s" % (what, CPD_ERRORS.get(rc, rc), extra))


def farr(v):
    return (ctypes.c_float * len(v))(*[float(x) for x in v])


def iarr(v):
    return (ctypes.c_int32 * len(v))(*[int(x) for x in v])


def ptr(t):
    """Device (or host) pointer of a contiguous tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "cpd_hip: tensor must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The current HIP stream of the current device as a raw handle (every launch takes it: the raw query is several
    times cheaper than building a torch.cuda.Stream object per call)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_SIDE_STREAMS = {}


def side_stream(device, role="index"):
    """The side stream that goes with the CURRENT stream of `device` for `role` ("index": the engines' index chain / deblock, the
    trainer's index chain; "wgrad": the trainer's weight gradients) -- one per (device, current stream, role) for the life of the
    process. Engines that run one after the other on the same stream share it. Why not a fresh torch.cuda.Stream() per engine: torch
    deals streams out of a pool and HIP maps them onto a few hardware queues in creation order, so every further stream raises the
    chance that a side stream lands on its own main stream's queue and serialises with it (measured in bench.py's extras, round 5:
    two 4-frame batches in flight 1088 -> 893 frames/s, the train step 9.3 -> 11.9 ms)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, int(torch.cuda.current_stream(dev).cuda_stream), role)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s
