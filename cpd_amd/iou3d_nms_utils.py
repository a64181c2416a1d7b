"""Mirror of cpd/ops/iou3d_nms/iou3d_nms_utils.py (same names, arguments and return conventions)
on top of the HIP extension. Differences are internal only: 3D IoU is one fused kernel instead of
overlap kernel + 8 torch ops, and NMS keeps its mask and greedy scan on the device."""
import numpy as np
import torch

from . import iou3d_nms_cuda, ops


def _check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:12-28."""
    boxes_a, is_numpy = _check_numpy_to_torch(boxes_a)
    boxes_b, is_numpy = _check_numpy_to_torch(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    assert boxes_a.shape[1] == 7 and boxes_b.shape[1] == 7
    ans_iou = boxes_a.new_zeros(torch.Size((boxes_a.shape[0], boxes_b.shape[0])))
    iou3d_nms_cuda.boxes_iou_bev_cpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou.numpy() if is_numpy else ans_iou


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:31-45."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_dis(boxes_a, boxes_b):
    """iou3d_nms_utils.py:47-64 (pure torch in the reference as well)."""
    d = boxes_a[:, None, 0:2] - boxes_b[None, :, 0:2]
    return torch.sqrt((d ** 2).sum(-1))


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:67-100."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return ops.boxes_iou3d(boxes_a, boxes_b)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:103-118. Returns (kept indices into `boxes` (device, int64), None)."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    keep = ops.nms(boxes, thresh)               # device indices; no CPU round trip of the mask
    return order[keep].contiguous(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:121-135."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous()
    keep = ops.nms(boxes, thresh, normal=True)
    return order[keep].contiguous(), None
