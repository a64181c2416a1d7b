"""Dataloader pre-filter on the device: the steps between the raw sweep and the
voxelizer, so the voxelizer consumes device points directly.

  mask_points_by_range      cpd/utils/common_utils.py:60-63 + data_processor.py:84-85 (boolean indexing)
  shuffle_points            data_processor.py:103-120 (np.random.permutation; the permutation is an input)
  points_in_boxes_gpu       cpd/ops/roiaware_pool3d/roiaware_pool3d_utils.py (points_in_boxes_gpu) ->
                            roiaware_pool3d_kernel.cu:313-336
  merge_sweeps              waymo_unsupervised_dataset.py:333-360 (get_frame: every sweep through its own pose and the
                            inverse of the current pose, points_rigid_transform l.192-202; intensity / last column zeroed)
  points_in_boxes_cpu       roiaware_pool3d_utils.points_in_boxes_cpu -> roiaware_pool3d.cpp:128-168 (MARGIN 1e-2)
  crop_boxes / place_prototype / sample_prototype
                            waymo_unsupervised_dataset.py:205-331 (sample_prototype_cpu: the box crop of the prototype sampler;
                            its pickle read and score / class bookkeeping stay on the host, the point work is on the device)
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


def points_in_boxes_cpu(points, boxes):
    """points (N, 3), boxes (K, 7) device tensors -> point_indices (K, N) int32, 1 where the point lies in the box: the
    reference's CPU test (MARGIN 1e-2, comparisons in double), here on the device."""
    assert boxes.shape[1] == 7 and points.shape[1] == 3
    points, boxes = points.contiguous().float(), boxes.contiguous().float()
    out = torch.zeros((boxes.shape[0], points.shape[0]), dtype=torch.int32, device=points.device)
    check(lib().cpd_points_in_boxes_mask(ptr(boxes), boxes.shape[0], ptr(points), points.shape[0], 3, ptr(out), stream()),
          "cpd_points_in_boxes_mask")
    return out


def crop_boxes(points, boxes, discard):
    """points [N, C], boxes [K, 7], discard [K] bool/int -> (points_no_obj, points_good_obj): the rows outside EVERY box and
    the rows outside every box flagged `discard`, order preserved (l.255-259, 317-318) -- one pass, no K x N matrix; one host
    read of the two counts."""
    points = points.contiguous().float()
    boxes = boxes.contiguous().float().reshape(-1, 7)
    n, c = points.shape
    k = boxes.shape[0]
    dis = torch.as_tensor(discard, device=points.device).to(torch.int32).contiguous()
    assert dis.numel() == k
    out_a, out_b = torch.empty_like(points), torch.empty_like(points)
    cnt = torch.zeros((2,), dtype=torch.int32, device=points.device)
    ws = torch.empty((lib().cpd_crop_boxes_workspace_bytes(n),), dtype=torch.uint8, device=points.device)
    check(lib().cpd_crop_boxes(ptr(points), n, c, ptr(boxes), ptr(dis), k, ptr(out_a), ptr(cnt[0:1]), ptr(out_b), ptr(cnt[1:2]),
                               ptr(ws), ws.numel(), stream()), "cpd_crop_boxes")
    na, nb = cnt.tolist()
    return out_a[:na], out_b[:nb]


def _pose_matrix(box):
    """trans_mat of l.281-289 / l.294-302, as the reference builds it: float32 identity with cos / sin (evaluated by numpy on the
    box's own dtype) and the centre written in."""
    x, y, z, yaw = box[0], box[1], box[2], box[6]
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = np.cos(yaw); m[0, 1] = -np.sin(yaw); m[0, 3] = x
    m[1, 0] = np.sin(yaw); m[1, 1] = np.cos(yaw); m[1, 3] = y
    m[2, 3] = z
    return m


def place_prototype(proto_points, proto_box, box, num_features):
    """One prototype instance moved into a target box (l.267-307): the prototype's points inside its own box (CPU in-box test),
    through the inverse of the prototype box's pose and then the target box's pose. proto_points [P, >=3] device tensor,
    proto_box / box host arrays of 7. Returns [P', num_features] fp32, xyz in columns 0-2, zeros elsewhere. The 4x4 matrices
    (and the float32 inverse) are formed on the host by the same numpy calls as the reference."""
    pp = proto_points[:, :3].contiguous().float()
    pb = np.asarray(proto_box)
    inside = points_in_boxes_cpu(pp, torch.as_tensor(np.asarray(pb[:7], np.float32)[None], device=pp.device))[0] != 0
    pp = pp[inside].contiguous()
    a = np.ascontiguousarray(np.linalg.inv(_pose_matrix(pb)).astype(np.float64))
    b = np.ascontiguousarray(_pose_matrix(np.asarray(box)).astype(np.float64))
    out = torch.empty((pp.shape[0], int(num_features)), dtype=torch.float32, device=pp.device)
    dp = ctypes.POINTER(ctypes.c_double)
    check(lib().cpd_transform_points(ptr(pp), pp.shape[0], 3, a.ctypes.data_as(dp), b.ctypes.data_as(dp), int(num_features), ptr(out),
                                     stream()), "cpd_transform_points")
    return out


def sample_prototype(points, outline_boxes, outline_cls, outline_score, proto_id, proto_points_set, discard_thresh_max,
                     discard_thresh_min, coin=0, permutation=None):
    """sample_prototype_cpu (waymo_unsupervised_dataset.py:205-331) with the prototype set passed in (the reference unpickles
    `<seq>_outline_<method>_CSS_proto.pkl`: {'proto_points_set': {class: {id: {'points', 'box'}}}}) and the two random draws as
    inputs (`coin` = np.random.randint(2), `permutation` = np.random.permutation(len(points_good_obj)) when the coin is 1).
    points [N, C] device tensor; boxes / names / scores / ids host arrays. Returns the reference's 6-tuple with the two clouds as
    device tensors: (points_good_obj, points_proto, new_outline_boxes, new_outline_cls, new_outline_score, new_proto_id)."""
    boxes = np.asarray(outline_boxes)
    k = len(boxes)
    discard = np.ones(k, bool)
    new_boxes, new_cls, new_score, new_id, protos = [], [], [], [], []
    for i in range(k):
        name, score, pid = outline_cls[i], outline_score[i], proto_id[i]
        if name in ("Vehicle", "Pedestrian", "Cyclist"):
            mx, mn = discard_thresh_max[name], discard_thresh_min[name]
            if score > min(mn, mx) and np.linalg.norm(boxes[i][0:2]) < 75 and pid >= 0:
                discard[i] = False
                new_boxes.append(boxes[i]); new_cls.append(name); new_id.append(pid)
                score = min(max(score, mn), mx)
                new_score.append((score - mn) / (mx - mn))
                proto = proto_points_set[name][pid]
                pts = proto["points"]
                pts = pts if torch.is_tensor(pts) else torch.as_tensor(np.asarray(pts, np.float32), device=points.device)
                protos.append(place_prototype(pts.to(points.device), proto["box"], boxes[i], points.shape[1]))
    bx = torch.as_tensor(np.asarray(boxes, np.float32).reshape(-1, boxes.shape[-1] if k else 7)[:, :7], device=points.device)
    no_obj, good = crop_boxes(points, bx, discard)
    points_proto = torch.cat(protos + [no_obj], 0)
    if coin:
        perm = torch.as_tensor(np.asarray(permutation), device=points.device).long()
        assert perm.numel() == good.shape[0]
        good = good.index_select(0, perm[:int(good.shape[0] * 0.2)])
    return good, points_proto, np.array(new_boxes), np.array(new_cls), np.array(new_score), np.array(new_id)
