"""Dataloader pre-filter on the device: the steps between the raw sweep and the
voxelizer, so the voxelizer consumes device points directly.

  mask_points_by_range      cpd/utils/common_utils.py:60-63 + data_processor.py:84-85 (boolean indexing)
  shuffle_points            data_processor.py:103-120 (np.random.permutation; the permutation is an input)
  points_in_boxes_gpu       cpd/ops/roiaware_pool3d/roiaware_pool3d_utils.py (points_in_boxes_gpu) ->
                            roiaware_pool3d_kernel.cu:313-336
  merge_sweeps              waymo_unsupervised_dataset.py:333-360 (get_frame: every sweep through its own pose and the
                            inverse of the current pose, points_rigid_transform l.192-202; intensity / last column zeroed)
"""
import ctypes

import numpy as np
import torch

from ._lib import check, farr, iarr, lib, ptr, stream


def mask_points_by_range(points, limit_range):
    """points [N, C] f32 device -> the rows inside the x/y range, order preserved (one host read of the count)."""
    points = points.contiguous()
    n, c = points.shape
    out = torch.empty_like(points)
    n_out = torch.zeros((1,), dtype=torch.int32, device=points.device)
    ws = torch.empty((lib().cpd_mask_points_workspace_bytes(n),), dtype=torch.uint8, device=points.device)
    check(lib().cpd_mask_points_by_range(ptr(points), n, c, farr(limit_range), ptr(out), ptr(n_out), ptr(ws), ws.numel(),
                                         stream()), "cpd_mask_points_by_range")
    return out[:int(n_out.item())]


def shuffle_points(points, permutation):
    """points[permutation] on the device; `permutation` plays np.random.permutation(N)'s role."""
    return points.index_select(0, permutation.to(points.device).long())


def points_in_boxes_gpu(points, boxes, margin=1e-5):
    """points (B, M, 3), boxes (B, T, 7) [x, y, z, dx, dy, dz, heading] -> box_idxs_of_pts (B, M) int32,
    -1 = background (first containing box otherwise)."""
    assert boxes.shape[0] == points.shape[0] and boxes.shape[2] == 7 and points.shape[2] == 3
    points, boxes = points.contiguous().float(), boxes.contiguous().float()
    b, m, _ = points.shape
    out = torch.empty((b, m), dtype=torch.int32, device=points.device)
    check(lib().cpd_points_in_boxes(b, boxes.shape[1], m, ptr(boxes), ptr(points), 3, float(margin), ptr(out), stream()),
          "cpd_points_in_boxes")
    return out


def merge_sweeps(sweeps, poses, cur_pose):
    """get_frame's multi-sweep merge on the device: `sweeps` = list of [N_i, C] f32 device tensors (oldest first), `poses` their
    4x4 sweep -> world matrices, `cur_pose` the current frame's pose (its inverse is taken here in float64 like the
    reference's np.linalg.inv). Returns the concatenated [sum N_i, C] cloud in the current frame."""
    assert len(sweeps) == len(poses) and len(sweeps) > 0
    pts = torch.cat([s.contiguous().float() for s in sweeps]) if len(sweeps) > 1 else sweeps[0].contiguous().float()
    offs = [0]
    for s in sweeps:
        offs.append(offs[-1] + s.shape[0])
    P = np.ascontiguousarray(np.stack([np.asarray(p, np.float64).reshape(4, 4) for p in poses]))
    inv = np.ascontiguousarray(np.linalg.inv(np.asarray(cur_pose, np.float64).reshape(4, 4)))
    out = torch.empty_like(pts)
    dp = ctypes.POINTER(ctypes.c_double)
    check(lib().cpd_merge_sweeps(ptr(pts), iarr(offs), len(sweeps), pts.shape[1], P.ctypes.data_as(dp), inv.ctypes.data_as(dp),
                                 ptr(out), stream()), "cpd_merge_sweeps")
    return out
