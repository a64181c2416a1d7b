"""The first stage of the dbscan / oyster two-stage configs executed on the CPU ORACLE (test infrastructure only):

  AnchorHeadSingleV2.forward              cpd/models/dense_heads/anchor_head_single.py:31-192   (shared conv, five get_layer branches l.9-29,
                                          direction classifier, occupancy anchor mask l.85-127, generate_predicted_boxes)
  AnchorHeadTemplate.generate_predicted_boxes   anchor_head_template.py:336-383                 (oracle.anchor_decode)
  RoIHeadTemplate.proposal_layer          cpd/models/roi_heads/roi_head_template.py:53-114      (max class score, class_agnostic_nms
                                          model_nms_utils.py:113-134 with NMS_CONFIG.TEST, labels + 1, zero-padded slots)

unfused and in NCHW like the reference modules: conv, bias, eval BatchNorm (eps 1e-5), ReLU one by one through oracle.conv2d /
oracle.bn_relu. The anchors are an INPUT here (cpd_amd.anchor_head.AnchorGenerator on the CPU, itself pinned on the reference class's
golden in tests/test_anchor_head.py)."""
import numpy as np

import ref_pipeline as rp

F = np.float32
BRANCHES = ("conv_cls", "conv_reg", "conv_height", "conv_dim", "conv_ang")


def anchor_mask(points_xy, voxel_size_x, pc_range, h, w):
    """get_anchor_mask (anchor_head_single.py:85-127): BEV cells within [-10, 10) cells of a coarse (x 10) cell holding a point of ANY frame of
    the batch; fp32 arithmetic, truncation toward zero, negative indices wrapping as torch indexing does."""
    stride = F(np.round(voxel_size_x * 8.0 * 10.0))
    x = ((points_xy[:, 0].astype(F) - F(pc_range[0])) / stride).astype(np.int64)
    y = ((points_xy[:, 1].astype(F) - F(pc_range[1])) / stride).astype(np.int64)
    x = np.minimum(x, w // 10 - 1)
    y = np.minimum(y, h // 10 - 1)
    large = np.zeros((h // 10, w // 10), np.int32)
    large[y, x] = 1
    idx = np.argwhere(large > 0) * 10
    mask = np.zeros((h, w), bool)
    for i in range(-10, 10):
        for j in range(-10, 10):
            mask[idx[:, 0] + i, idx[:, 1] + j] = True
    return mask


def head(o, sd, bev, prefix="dense_head."):
    """-> (cls (B, A nc, H, W), box (B, 7 A, H, W), dir (B, 2 A, H, W) | None)"""
    p = prefix
    shard = o.conv2d(bev, rp._np(sd, p + "shared_conv.0.weight"), rp._np(sd, p + "shared_conv.0.bias"), 1, 1)
    shard = rp.bn2d(o, shard, sd, p + "shared_conv.1", 1e-5)
    outs = {}
    for br in BRANCHES:
        q = p + br
        hdn = o.conv2d(shard, rp._np(sd, q + ".0.weight"), rp._np(sd, q + ".0.bias"), 1, 1)
        hdn = rp.bn2d(o, hdn, sd, q + ".1", 1e-5)
        outs[br] = o.conv2d(hdn, rp._np(sd, q + ".3.weight"), rp._np(sd, q + ".3.bias"), 1, 0)
    box = np.concatenate([outs["conv_reg"], outs["conv_height"], outs["conv_dim"], outs["conv_ang"]], 1)
    dirs = None
    if p + "conv_dir_cls.weight" in sd:
        dirs = o.conv2d(bev, rp._np(sd, p + "conv_dir_cls.weight"), rp._np(sd, p + "conv_dir_cls.bias"), 1, 0)
    return outs["conv_cls"], box, dirs


def predicted_boxes(o, head_cfg, anchors_root, mask, cls, box, dirs, num_class):
    """forward's tail: masked anchors / predictions (l.129-190) and generate_predicted_boxes. anchors_root: list per class of
    (1, H, W, n_size, n_rot, 7) arrays. -> (batch_cls_preds (B, N, nc), batch_box_preds (B, N, 7), anchors (N, 7))"""
    b = cls.shape[0]
    anchors = np.concatenate([np.asarray(a, F)[:, mask] for a in anchors_root], axis=-3)          # (1, K, n_cls, n_rot, 7)
    flat = np.ascontiguousarray(anchors.reshape(-1, 7))
    n = flat.shape[0]
    pick = lambda t: np.ascontiguousarray(np.transpose(t, (0, 2, 3, 1))[:, mask, :])
    cls_p = pick(cls).reshape(b, n, num_class)
    box_p = pick(box).reshape(b, n, 7)
    dir_p = pick(dirs).reshape(b, n, -1) if dirs is not None else None
    dec = o.anchor_decode(box_p, flat, dir_p, head_cfg.get("DIR_OFFSET", 0.78539), head_cfg.get("DIR_LIMIT_OFFSET", 0.0))
    return cls_p.astype(F), dec, flat


def proposal_layer(o, nms_cfg, batch_box_preds, batch_cls_preds):
    """roi_head_template.py:53-114 with MULTI_CLASSES_NMS False: per frame the max class score, the NMS_PRE_MAXSIZE best in descending
    order (ties -> lower index), rotated NMS at NMS_THRESH, the first NMS_POST_MAXSIZE; labels = argmax + 1 on EVERY slot of the zero
    buffer (l.111)."""
    b = batch_box_preds.shape[0]
    post, pre = int(nms_cfg["NMS_POST_MAXSIZE"]), int(nms_cfg["NMS_PRE_MAXSIZE"])
    rois = np.zeros((b, post, 7), F)
    scores = np.zeros((b, post), F)
    labels = np.zeros((b, post), np.int64)
    kept, ranked = [], []
    for i in range(b):
        s, lab = batch_cls_preds[i].max(-1), batch_cls_preds[i].argmax(-1)
        order = np.argsort(-s, kind="stable")[:pre]
        keep = o.nms(np.ascontiguousarray(batch_box_preds[i][order].astype(F)), float(nms_cfg["NMS_THRESH"]))
        sel = order[keep][:post]
        rois[i, :len(sel)] = batch_box_preds[i][sel]
        scores[i, :len(sel)] = s[sel]
        labels[i, :len(sel)] = lab[sel]
        kept.append(len(sel))
        ranked.append(order)
    return rois, scores, labels + 1, kept, ranked


def first_stage(o, cfg, head_cfg, nms_cfg, sd, points, anchors_root, mask_points_xy=None):
    """One frame through the oracle: voxelizer ... BaseBEVBackbone (ref_pipeline), then the anchor head and the proposal layer.
    `mask_points_xy`: the xy of every point of the BATCH the frame travels in (the reference builds ONE occupancy mask per batch)."""
    feats, coords = rp.voxelize_batch(o, cfg, [points])
    levels, (x, idx, shape) = rp.backbone3d(o, cfg, sd, feats, coords, 1)
    spatial = o.densify(x, idx, 1, shape)
    bev = rp.bev_backbone(o, cfg, sd, spatial)
    cls, box, dirs = head(o, sd, bev)
    h, w = bev.shape[-2:]
    vx = (cfg.point_cloud_range[3] - cfg.point_cloud_range[0]) / float(round((cfg.point_cloud_range[3] - cfg.point_cloud_range[0]) / cfg.voxel_size[0]))
    mask = anchor_mask(points[:, :2] if mask_points_xy is None else mask_points_xy, vx, cfg.point_cloud_range, h, w)
    cls_p, boxes, anchors = predicted_boxes(o, head_cfg, anchors_root, mask, cls, box, dirs, cfg.num_class)
    rois, scores, labels, kept, ranked = proposal_layer(o, nms_cfg, boxes, cls_p)
    return dict(rois=rois, roi_scores=scores, roi_labels=labels, kept=kept, ranked=ranked, levels=levels, bev=bev, mask=mask,
                batch_cls_preds=cls_p, batch_box_preds=boxes, anchors=anchors, head=(cls, box, dirs))
