"""The public names of cpd/ops/iou3d_nms/iou3d_nms_utils.py (boxes_bev_iou_cpu, boxes_iou_bev,
boxes_dis, boxes_iou3d_gpu, nms_gpu, nms_normal_gpu; same arguments and return conventions) served by
the HIP kernels of csrc/iou3d_nms.hip through cpd_amd.ops. Internally different from the reference
wrapper: 3D IoU is one fused kernel (not an overlap kernel plus eight torch ops), and NMS keeps the
suppression mask and the greedy scan on the device, so no N x N/64 mask crosses PCIe."""
import numpy as np
import torch

from . import ops


def _require_boxes(*tensors):
    for t in tensors:
        if t.dim() != 2 or t.shape[1] != 7:
            raise AssertionError("boxes must be (N, 7) [x, y, z, dx, dy, dz, heading], got %s" % (tuple(t.shape),))


def _descending(scores, limit=None):
    """Indices of `scores` from best to worst (the reference sorts with scores.sort(0, descending=True))."""
    idx = torch.argsort(scores, dim=0, descending=True)
    return idx if limit is None else idx[:limit]


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """Rotated BEV IoU on HOST data (iou3d_nms_utils.py:12-28; numpy in -> numpy out)."""
    was_numpy = isinstance(boxes_a, np.ndarray)
    a = torch.from_numpy(boxes_a).float() if isinstance(boxes_a, np.ndarray) else boxes_a
    b = torch.from_numpy(boxes_b).float() if isinstance(boxes_b, np.ndarray) else boxes_b
    if a.is_cuda or b.is_cuda:
        raise AssertionError("Only support CPU tensors")
    _require_boxes(a, b)
    iou = ops.boxes_iou_bev_cpu(a.contiguous(), b.contiguous())
    return iou.numpy() if was_numpy else iou


def boxes_iou_bev(boxes_a, boxes_b):
    """(N, M) rotated BEV IoU on the device (iou3d_nms_utils.py:31-45)."""
    _require_boxes(boxes_a, boxes_b)
    return ops.boxes_iou_bev(boxes_a, boxes_b)


def boxes_dis(boxes_a, boxes_b):
    """(N, M) centre distance in the xy plane (iou3d_nms_utils.py:47-64; torch in the reference too)."""
    return torch.cdist(boxes_a[:, 0:2].float(), boxes_b[:, 0:2].float())


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N, M) 3D IoU = BEV overlap x height overlap / union volume (iou3d_nms_utils.py:67-100)."""
    _require_boxes(boxes_a, boxes_b)
    return ops.boxes_iou3d(boxes_a, boxes_b)


def _nms(boxes, scores, thresh, limit, normal):
    _require_boxes(boxes)
    ranked = _descending(scores, limit)
    kept = ops.nms(boxes.index_select(0, ranked).contiguous(), thresh, normal=normal)    # positions in the ranked list
    return ranked.index_select(0, kept).contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated NMS (iou3d_nms_utils.py:103-118): (indices into `boxes` of the kept ones, best first; None)."""
    return _nms(boxes, scores, thresh, pre_maxsize, normal=False)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned NMS (iou3d_nms_utils.py:121-135)."""
    return _nms(boxes, scores, thresh, None, normal=True)
