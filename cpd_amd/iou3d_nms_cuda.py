"""Drop-in for the reference's compiled extension module `iou3d_nms_cuda`
(cpd/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17): same five entry points, same argument order,
caller-allocated outputs, integer return values. Contract violations raise instead of the
reference's fprintf + exit(-1) (iou3d_nms.cpp:14-26)."""
import torch

from . import _lib, ops
from ._lib import check, lib, ptr, stream


def _check_dev(t, name):
    if not t.is_cuda:
        raise _lib.CpdHipError("%s must be a CUDA/HIP tensor" % name)
    if not t.is_contiguous():
        raise _lib.CpdHipError("%s must be contiguous" % name)
    if t.dtype != torch.float32:
        raise _lib.CpdHipError("%s must be float32" % name)


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_nms.cpp:49-68: ans_overlap (N, M) filled in place; returns 1."""
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_overlap, "ans_overlap")):
        _check_dev(t, n)
    check(lib().cpd_boxes_overlap_bev(ptr(boxes_a), boxes_a.shape[0], ptr(boxes_b), boxes_b.shape[0], ptr(ans_overlap),
                                      stream()), "boxes_overlap_bev_gpu")
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    """iou3d_nms.cpp:70-88."""
    for t, n in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_iou, "ans_iou")):
        _check_dev(t, n)
    check(lib().cpd_boxes_iou_bev(ptr(boxes_a), boxes_a.shape[0], ptr(boxes_b), boxes_b.shape[0], ptr(ans_iou), stream()),
          "boxes_iou_bev_gpu")
    return 1


def _nms(boxes, keep, thresh, normal):
    _check_dev(boxes, "boxes")
    if keep.is_cuda or keep.dtype != torch.int64 or not keep.is_contiguous():
        raise _lib.CpdHipError("keep must be a contiguous CPU LongTensor (iou3d_nms_utils.py:116)")
    k_dev, num = ops.nms(boxes, thresh, normal=normal, sync=False)
    n = int(num.item())                       # synchronous, like the reference's cudaMemcpy
    keep[:n] = k_dev[:n].cpu()                # D2H of n indices instead of the N x N/64 mask
    return n


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:90-137: boxes (N,7) device, sorted by descending score; keep: CPU LongTensor(N);
    returns num_to_keep."""
    return _nms(boxes, keep, nms_overlap_thresh, False)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:139-186."""
    return _nms(boxes, keep, nms_overlap_thresh, True)


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:232-252: CPU tensors."""
    for t in (boxes_a, boxes_b, ans_iou):
        if t.is_cuda or not t.is_contiguous():
            raise _lib.CpdHipError("boxes_iou_bev_cpu wants contiguous CPU tensors")
    check(lib().cpd_boxes_iou_bev_cpu(ptr(boxes_a), boxes_a.shape[0], ptr(boxes_b), boxes_b.shape[0], ptr(ans_iou)),
          "boxes_iou_bev_cpu")
    return 1
