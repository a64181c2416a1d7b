"""Anchor head pieces on the C-ABI kernels of csrc/roi_pool.hip. Mirrors

  AnchorGenerator.generate_anchors            cpd/models/dense_heads/target_assigner/anchor_generator.py:18-62
  box_utils.boxes3d_nearest_bev_iou           cpd/utils/box_utils.py:275-287
  AxisAlignedTargetAssigner.assign_targets    cpd/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:46-243
  AnchorHeadTemplate.generate_predicted_boxes cpd/models/dense_heads/anchor_head_template.py:336-383
  AnchorHeadTemplate.get_loss                 cpd/models/dense_heads/anchor_head_template.py:179-334 (+ cpd/utils/loss_utils.py:10-206)

The assigner never materialises the anchors x GT IoU matrix: `cpd_anchor_assign` keeps the per-anchor
max / argmax and the per-GT max on chip (212k anchors x 100 GT boxes would be 85 MB per sample and class)."""
import math

import torch
import torch.nn.functional as F

from . import ops
from ._lib import check, farr, lib, ptr, stream


class AnchorGenerator:
    def __init__(self, anchor_range, anchor_generator_config):
        self.anchor_range = [float(v) for v in anchor_range]
        self.anchor_sizes = [c["anchor_sizes"] for c in anchor_generator_config]
        self.anchor_rotations = [c["anchor_rotations"] for c in anchor_generator_config]
        self.anchor_heights = [c["anchor_bottom_heights"] for c in anchor_generator_config]
        self.align_center = [c.get("align_center", False) for c in anchor_generator_config]
        self.num_of_anchor_sets = len(self.anchor_sizes)

    def generate_anchors(self, grid_sizes, device="cuda"):
        """-> ([ (1, H, W, n_size, n_rot, 7) per class ], [anchors per location per class])."""
        r = self.anchor_range
        all_anchors, per_loc = [], []
        for grid_size, sizes, rots, heights, center in zip(grid_sizes, self.anchor_sizes, self.anchor_rotations, self.anchor_heights,
                                                           self.align_center):
            per_loc.append(len(rots) * len(sizes) * len(heights))
            if center:
                xs, ys = (r[3] - r[0]) / grid_size[0], (r[4] - r[1]) / grid_size[1]
                xo, yo = xs / 2, ys / 2
            else:
                xs, ys = (r[3] - r[0]) / (grid_size[0] - 1), (r[4] - r[1]) / (grid_size[1] - 1)
                xo, yo = 0, 0
            x = torch.arange(r[0] + xo, r[3] + 1e-5, step=xs, dtype=torch.float32, device=device)
            y = torch.arange(r[1] + yo, r[4] + 1e-5, step=ys, dtype=torch.float32, device=device)
            z = x.new_tensor(heights)
            size, rot = x.new_tensor(sizes), x.new_tensor(rots)
            gx, gy, gz = torch.meshgrid([x, y, z], indexing="ij")
            a = torch.stack((gx, gy, gz), dim=-1)[:, :, :, None, :].repeat(1, 1, 1, size.shape[0], 1)
            a = torch.cat((a, size.view(1, 1, 1, -1, 3).repeat([*a.shape[0:3], 1, 1])), dim=-1)
            a = a[:, :, :, :, None, :].repeat(1, 1, 1, 1, rot.shape[0], 1)
            a = torch.cat((a, rot.view(1, 1, 1, 1, -1, 1).repeat([*a.shape[0:3], size.shape[0], 1, 1])), dim=-1)
            a = a.permute(2, 1, 0, 3, 4, 5).contiguous()
            a[..., 2] += a[..., 5] / 2
            all_anchors.append(a)
        return all_anchors, per_loc


def boxes3d_nearest_bev_iou(boxes_a, boxes_b):
    a, b = boxes_a[:, :7].contiguous().float(), boxes_b[:, :7].contiguous().float()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(lib().cpd_nearest_bev_iou(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out), stream()), "cpd_nearest_bev_iou")
    return out


def assign_targets_single(anchors, gt_boxes, gt_classes, matched_threshold=0.6, unmatched_threshold=0.45, norm_by_num_examples=False):
    """anchors [N,7], gt_boxes [M,7], gt_classes [M] -> dict like the reference's assign_targets_single."""
    anchors = anchors.contiguous().float()
    gt = gt_boxes[:, :7].contiguous().float()
    cls = gt_classes.int().contiguous()
    n, m = anchors.shape[0], gt.shape[0]
    dev = anchors.device
    labels = torch.empty((n,), dtype=torch.int32, device=dev)
    targets = torch.empty((n, 7), dtype=torch.float32, device=dev)
    weights = torch.empty((n,), dtype=torch.float32, device=dev)
    ious = torch.empty((n,), dtype=torch.float32, device=dev)
    ws = torch.empty((lib().cpd_anchor_assign_workspace_bytes(n, m),), dtype=torch.uint8, device=dev)
    check(lib().cpd_anchor_assign(ptr(anchors), n, ptr(gt) if m else None, m, ptr(cls) if m else None, float(matched_threshold),
                                  float(unmatched_threshold), int(bool(norm_by_num_examples)), ptr(labels), ptr(targets),
                                  ptr(weights), ptr(ious), ptr(ws), ws.numel(), stream()), "cpd_anchor_assign")
    return {"box_cls_labels": labels, "box_reg_targets": targets, "reg_weights": weights, "gt_ious": ious}


class AxisAlignedTargetAssigner:
    """assign_targets with the reference's output layout (use_multihead = False, match_height = False, no sampling)."""

    def __init__(self, anchor_generator_cfg, class_names, norm_by_num_examples=False):
        self.class_names = list(class_names)
        self.anchor_class_names = [c["class_name"] for c in anchor_generator_cfg]
        self.matched = {c["class_name"]: c["matched_threshold"] for c in anchor_generator_cfg}
        self.unmatched = {c["class_name"]: c["unmatched_threshold"] for c in anchor_generator_cfg}
        self.norm_by_num_examples = norm_by_num_examples

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        out = {"box_cls_labels": [], "box_reg_targets": [], "reg_weights": [], "gt_ious": []}
        gt_classes = gt_boxes_with_classes[:, :, -1]
        gt_boxes = gt_boxes_with_classes[:, :, :-1]
        for k in range(gt_boxes_with_classes.shape[0]):
            cur = gt_boxes[k]
            nz = (cur.abs().sum(1) != 0).nonzero()                      # trailing all-zero rows are padding (l.66-69)
            cnt = int(nz[-1]) + 1 if nz.numel() else 1
            cur, cur_cls = cur[:cnt], gt_classes[k][:cnt].int()
            per_class = []
            for name, anchors in zip(self.anchor_class_names, all_anchors):
                mask = torch.tensor([self.class_names[int(c) - 1] == name for c in cur_cls.tolist()], dtype=torch.bool,
                                    device=cur.device)
                fms = anchors.shape[:2]                                  # the reference's feature_map_size (l.92)
                t = assign_targets_single(anchors.view(-1, anchors.shape[-1]), cur[mask], cur_cls[mask], self.matched[name],
                                          self.unmatched[name], self.norm_by_num_examples)
                per_class.append((t, fms))
            out["box_reg_targets"].append(torch.cat([t["box_reg_targets"].view(*f, -1, 7) for t, f in per_class], dim=-2).view(-1, 7))
            for key in ("box_cls_labels", "gt_ious", "reg_weights"):
                out[key].append(torch.cat([t[key].view(*f, -1) for t, f in per_class], dim=-1).view(-1))
        return {k: torch.stack(v, dim=0) for k, v in out.items()}


class ATSSTargetAssigner:
    """ATSSTargetAssigner (atss_target_assigner.py:8-141) with the reference's constructor and `assign_targets` contract:
    anchors_list = one tensor or a list of per-class anchor tensors, gt_boxes_with_classes (B, M, 8); every anchor set is
    matched against ALL boxes of a frame (no class filter, as in the reference); returns box_cls_labels (B, N) float class
    ids, box_reg_targets (B, N, 7), reg_weights (B, N). One `cpd_atss_assign` call per (frame, anchor set); no N x M
    matrices. box_coder is accepted for signature compatibility (ResidualCoder, code_size 7, is what the kernel encodes)."""

    def __init__(self, topk, box_coder=None, match_height=False, **kwargs):
        # **kwargs: AnchorHeadTemplate.get_target_assigner (anchor_head_template.py:64-69) passes use_multihead= here, which the
        # reference's own constructor does not take (its ATSS branch raises TypeError as shipped); accepted and ignored
        self.topk = int(topk)
        self.box_coder = box_coder
        self.match_height = bool(match_height)
        assert box_coder is None or getattr(box_coder, "code_size", 7) == 7

    def assign_targets(self, anchors_list, gt_boxes_with_classes, use_multihead=False):
        if not isinstance(anchors_list, list):
            anchors_list = [anchors_list]
        gt = gt_boxes_with_classes.contiguous().float()
        B, M, ld = gt.shape
        assert ld >= 8
        # trailing all-zero boxes are padding, at least one row stays (l.40-45): one host read for the whole batch
        nz = (gt[:, :, :-1].sum(-1) != 0)
        last = torch.where(nz.any(1), M - 1 - nz.flip(1).float().argmax(1), torch.zeros(B, device=gt.device, dtype=torch.long))
        counts = (last + 1).tolist()
        outs = {"box_cls_labels": [], "box_reg_targets": [], "reg_weights": []}
        for anchors in anchors_list:
            if use_multihead:
                a = anchors.permute(3, 4, 0, 1, 2, 5).contiguous().view(-1, anchors.shape[-1])
            else:
                a = anchors.reshape(-1, anchors.shape[-1])
            a = a[:, :7].contiguous().float()
            n = a.shape[0]
            labels = torch.empty((B, n), dtype=torch.float32, device=gt.device)
            targets = torch.empty((B, n, 7), dtype=torch.float32, device=gt.device)
            weights = torch.empty((B, n), dtype=torch.float32, device=gt.device)
            for b in range(B):
                m = int(counts[b])
                ws = torch.empty((lib().cpd_atss_workspace_bytes(m, self.topk),), dtype=torch.uint8, device=gt.device)
                check(lib().cpd_atss_assign(ptr(a), n, ptr(gt[b]), ld, m, self.topk, int(self.match_height), ptr(labels[b]),
                                            ptr(targets[b]), ptr(weights[b]), ptr(ws), ws.numel(), stream()), "cpd_atss_assign")
            outs["box_cls_labels"].append(labels)
            outs["box_reg_targets"].append(targets)
            outs["reg_weights"].append(weights)
        return {k: (v[0] if len(v) == 1 else torch.cat(v, dim=1)) for k, v in outs.items()}


def get_target_assigner(anchor_target_cfg, anchor_generator_cfg=None, class_names=None, box_coder=None):
    """AnchorHeadTemplate.get_target_assigner (anchor_head_template.py:62-81): TARGET_ASSIGNER_CONFIG.NAME selects the class."""
    get = (lambda k, d=None: anchor_target_cfg.get(k, d)) if hasattr(anchor_target_cfg, "get") else (lambda k, d=None: getattr(anchor_target_cfg, k, d))
    name = get("NAME")
    if name == "ATSS":
        return ATSSTargetAssigner(topk=get("TOPK"), box_coder=box_coder, match_height=bool(get("MATCH_HEIGHT", False)))
    if name == "AxisAlignedTargetAssigner":
        return AxisAlignedTargetAssigner(anchor_generator_cfg, class_names, norm_by_num_examples=bool(get("NORM_BY_NUM_EXAMPLES", False)))
    raise NotImplementedError(name)


def generate_predicted_boxes(anchors, batch_size, cls_preds, box_preds, dir_cls_preds=None, dir_offset=0.78539,
                             dir_limit_offset=0.0, num_dir_bins=2):
    """anchors: list of per-class anchor tensors (or one tensor); cls/box/dir preds (B, H, W, C*). Returns
    (batch_cls_preds (B, N, num_class), batch_box_preds (B, N, 7))."""
    if isinstance(anchors, list):
        anchors = torch.cat(anchors, dim=-3)
    a = anchors.reshape(-1, anchors.shape[-1]).contiguous().float()
    n = a.shape[0]
    bp = box_preds.reshape(batch_size, n, -1).contiguous().float()
    assert bp.shape[-1] == 7
    dc = dir_cls_preds.reshape(batch_size, n, -1).contiguous().float() if dir_cls_preds is not None else None
    out = torch.empty_like(bp)
    check(lib().cpd_anchor_decode(ptr(bp), ptr(a), ptr(dc), batch_size, n, dc.shape[-1] if dc is not None else 0, float(dir_offset),
                                  float(dir_limit_offset), ptr(out), stream()), "cpd_anchor_decode")
    return cls_preds.reshape(batch_size, n, -1).float(), out


def _flat_anchors(anchors):
    if isinstance(anchors, list):
        anchors = torch.cat(anchors, dim=-3)
    return anchors.reshape(-1, anchors.shape[-1]).contiguous().float()


def anchor_head_loss_torch(anchors, cls_preds, box_preds, dir_cls_preds, box_cls_labels, box_reg_targets, num_class,
                           cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=None, dir_offset=0.78539, num_dir_bins=2):
    """Torch restatement of get_cls_layer_loss + get_box_reg_layer_loss (autograd-capable, any device): the form the fused
    kernel is tested against, itself pinned on tests/golden/anchor_loss.npz (the reference's own get_loss and gradients).
    Returns (total, {"rpn_loss_cls", "rpn_loss_loc", "rpn_loss_dir"})."""
    a = _flat_anchors(anchors).to(cls_preds.device)
    n = a.shape[0]
    B = cls_preds.shape[0]
    labels = box_cls_labels.reshape(B, n).long()
    pos = labels > 0
    norm = pos.sum(1, keepdim=True).float().clamp(min=1.0)
    # classification: focal-weighted sigmoid cross entropy against the one-hot class (all zeros for background / ignored)
    x = cls_preds.reshape(B, n, num_class).float()
    t = F.one_hot(labels.clamp(min=0), num_class + 1)[..., 1:].to(x.dtype)
    pr = torch.sigmoid(x)
    focal = (t * 0.25 + (1 - t) * 0.75) * (t * (1 - pr) + (1 - t) * pr) ** 2
    bce = x.clamp(min=0) - x * t + torch.log1p(torch.exp(-x.abs()))
    w_cls = (labels >= 0).float() / norm
    cls_loss = (focal * bce * w_cls.unsqueeze(-1)).sum() / B * cls_weight
    # box regression: heading enters as sin(pred - target), written out so that both factors carry gradient
    bp = box_preds.reshape(B, n, 7).float()
    rt = box_reg_targets.reshape(B, n, 7).float()
    enc_p = torch.cat([bp[..., :6], torch.sin(bp[..., 6:7]) * torch.cos(rt[..., 6:7])], -1)
    enc_t = torch.cat([rt[..., :6], torch.cos(bp[..., 6:7]) * torch.sin(rt[..., 6:7])], -1)
    enc_t = torch.where(torch.isnan(enc_t), enc_p, enc_t)
    cw = bp.new_tensor(code_weights if code_weights is not None else [1.0] * 7)
    d = ((enc_p - enc_t) * cw).abs()
    beta = 1.0 / 9.0
    w_reg = pos.float() / norm
    loc_loss = (torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta) * w_reg.unsqueeze(-1)).sum() / B * loc_weight
    parts = {"rpn_loss_cls": cls_loss.detach(), "rpn_loss_loc": loc_loss.detach()}
    total = cls_loss + loc_loss
    if dir_cls_preds is not None:
        rot = rt[..., 6] + a[:, 6].unsqueeze(0)
        v = rot - dir_offset
        off = v - torch.floor(v / (2 * math.pi)) * (2 * math.pi)
        bins = torch.floor(off / (2 * math.pi / num_dir_bins)).long().clamp(0, num_dir_bins - 1)
        logits = dir_cls_preds.reshape(B, n, num_dir_bins).float()
        ce = F.cross_entropy(logits.reshape(-1, num_dir_bins), bins.reshape(-1), reduction="none").view(B, n)
        dir_loss = (ce * w_reg).sum() / B * dir_weight
        parts["rpn_loss_dir"] = dir_loss.detach()
        total = total + dir_loss
    return total, parts


def anchor_head_loss(anchors, cls_preds, box_preds, dir_cls_preds, box_cls_labels, box_reg_targets, num_class,
                     cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=None, dir_offset=0.78539, num_dir_bins=2):
    """get_loss fused with its gradient on the device (cpd_anchor_loss). Returns (losses[4] = total, cls, loc, dir;
    (d_cls_preds, d_box_preds, d_dir_cls_preds)) with the gradients shaped like the predictions."""
    a = _flat_anchors(anchors)
    n = a.shape[0]
    B = cls_preds.shape[0]
    cp = cls_preds.reshape(B, n, num_class).contiguous().float()
    bp = box_preds.reshape(B, n, 7).contiguous().float()
    dp = dir_cls_preds.reshape(B, n, num_dir_bins).contiguous().float() if dir_cls_preds is not None else None
    lab = box_cls_labels.reshape(B, n).to(torch.int32).contiguous()
    rt = box_reg_targets.reshape(B, n, 7).contiguous().float()
    d_cls, d_box = torch.empty_like(cp), torch.empty_like(bp)
    d_dir = torch.empty_like(dp) if dp is not None else None
    losses = torch.empty(4, dtype=torch.float32, device=cp.device)
    ws = torch.empty(lib().cpd_anchor_loss_workspace_bytes(B, n), dtype=torch.uint8, device=cp.device)
    check(lib().cpd_anchor_loss(ptr(cp), ptr(bp), ptr(dp), ptr(lab), ptr(rt), ptr(a), B, n, num_class, num_dir_bins, float(dir_offset),
                                farr(code_weights if code_weights is not None else [1.0] * 7), float(cls_weight), float(loc_weight),
                                float(dir_weight), ptr(d_cls), ptr(d_box), ptr(d_dir), ptr(losses), ptr(ws), ws.numel(), stream()),
          "cpd_anchor_loss")
    return losses, (d_cls.view_as(cls_preds), d_box.view_as(box_preds), d_dir.view_as(dir_cls_preds) if d_dir is not None else None)


class AnchorHeadSingle(torch.nn.Module):
    """Inference path of AnchorHeadSingle (cpd/models/dense_heads/anchor_head_single.py:194-356) + get_loss: occupancy anchor
    mask, the 1x1 conv_cls / conv_box / conv_dir_cls (cpd_gather_conv through cpd_amd.models.Conv2d; state_dict
    names as in the reference), masked anchors, generate_predicted_boxes (cpd_anchor_decode)."""

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range,
                 predict_boxes_when_training=True, **kwargs):
        super().__init__()
        from .models import Conv2d
        self.model_cfg, self.num_class, self.class_names = model_cfg, num_class, list(class_names)
        self.predict_boxes_when_training = bool(predict_boxes_when_training)   # ctor arg as in the reference (l.195-196), default True
        self.range = [float(v) for v in point_cloud_range]
        self.voxel_size = (self.range[3] - self.range[0]) / float(grid_size[0])
        agc = model_cfg["ANCHOR_GENERATOR_CONFIG"]
        fms = [[int(grid_size[0]) // c["feature_map_stride"], int(grid_size[1]) // c["feature_map_stride"]] for c in agc]
        self._gen = (AnchorGenerator(self.range, agc), fms)
        self.anchors_root = None                                     # built on the first forward (device known then)
        self.anchors = None                                          # the occupancy-masked anchors of the last forward
        self.forward_ret_dict = {}
        per_loc = sum(len(c["anchor_rotations"]) * len(c["anchor_sizes"]) * len(c["anchor_bottom_heights"]) for c in agc)
        self.num_anchors_per_location = per_loc
        self.conv_cls = Conv2d(input_channels, per_loc * num_class, kernel_size=1)
        self.conv_box = Conv2d(input_channels, per_loc * 7, kernel_size=1)
        self.num_dir_bins = model_cfg.get("NUM_DIR_BINS", 2)
        self.conv_dir_cls = Conv2d(input_channels, per_loc * self.num_dir_bins, kernel_size=1) \
            if model_cfg.get("USE_DIRECTION_CLASSIFIER", None) is not None else None

    def get_anchor_mask(self, points, shape):
        """l.238-278: BEV cells within +-10 cells of a coarse (x10) cell that contains a point. The reference builds
        the index list in numpy and relies on torch indexing, negative indices wrapping around included."""
        h, w = shape[-2], shape[-1]
        stride = float(torch.round(torch.tensor(self.voxel_size * 8.0 * 10.0)))
        dev = points.device
        in_x = ((points[:, 1] - self.range[0]) / stride).long().clamp(max=w // 10 - 1)
        in_y = ((points[:, 2] - self.range[1]) / stride).long().clamp(max=h // 10 - 1)
        coarse = torch.zeros(h // 10, w // 10, device=dev)
        coarse[in_y, in_x] = 1
        idx = coarse.nonzero() * 10                                  # (K, 2) row, col
        off = torch.arange(-10, 10, device=dev)
        rows = (idx[:, None, None, 0] + off[None, :, None]).expand(-1, 20, 20).reshape(-1)
        cols = (idx[:, None, None, 1] + off[None, None, :]).expand(-1, 20, 20).reshape(-1)
        mask = torch.zeros(h, w, device=dev)
        mask[rows, cols] = 1
        return mask.bool()

    @property
    def target_assigner(self):
        """AnchorHeadTemplate.get_target_assigner (anchor_head_template.py:62-81), built on first use."""
        if getattr(self, "_assigner", None) is None:
            self._assigner = get_target_assigner(self.model_cfg["TARGET_ASSIGNER_CONFIG"], self.model_cfg["ANCHOR_GENERATOR_CONFIG"],
                                                 self.class_names)
        return self._assigner

    def assign_targets(self, gt_boxes):
        """AnchorHeadTemplate.assign_targets (anchor_head_template.py:116-127): the assigner on the anchors of THIS forward (the
        occupancy-masked ones) and gt_boxes (B, M, 8)."""
        return self.target_assigner.assign_targets(self.anchors, gt_boxes)

    def _train_targets(self, data_dict):
        """anchor_head_single.py:176-181 / 335-340: in training mode the targets of this batch join forward_ret_dict and the
        per-anchor IoUs go to data_dict['gt_ious'] (the axis-aligned assigner provides them; part_wraper.py:137 reads them)."""
        targets = self.assign_targets(data_dict["gt_boxes"])
        self.forward_ret_dict.update(targets)
        if "gt_ious" in targets:
            data_dict["gt_ious"] = targets["gt_ious"]

    def get_loss(self, forward_ret_dict=None, anchors=None):
        """AnchorHeadTemplate.get_loss (anchor_head_template.py:321-334) for a forward_ret_dict with the reference's keys
        (cls_preds, box_preds, dir_cls_preds, box_cls_labels, box_reg_targets); defaults: the dict and the (masked) anchors of
        the last forward -- the predictions are laid out on those. Fused loss + gradient kernel; returns
        (losses[4] = rpn_loss, cls, loc, dir on the device, (d_cls_preds, d_box_preds, d_dir_cls_preds))."""
        lw = self.model_cfg["LOSS_CONFIG"]["LOSS_WEIGHTS"]
        f = forward_ret_dict if forward_ret_dict is not None else self.forward_ret_dict
        if anchors is None:
            anchors = self.anchors if self.anchors is not None else self.anchors_root
        return anchor_head_loss(anchors, f["cls_preds"], f["box_preds"],
                                f.get("dir_cls_preds"), f["box_cls_labels"], f["box_reg_targets"], self.num_class,
                                lw["cls_weight"], lw["loc_weight"], lw.get("dir_weight", 0.2), lw["code_weights"],
                                self.model_cfg.get("DIR_OFFSET", 0.78539), self.num_dir_bins)

    def _predict(self, data_dict, cls_preds, box_preds, dir_preds):
        """generate_predicted_boxes on detached predictions (eval, or training with predict_boxes_when_training: l.183-190)."""
        with torch.no_grad():
            cls, boxes = generate_predicted_boxes(self.anchors, data_dict["batch_size"], cls_preds.detach(), box_preds.detach(),
                                                  dir_preds.detach() if dir_preds is not None else None,
                                                  self.model_cfg.get("DIR_OFFSET", 0.78539), self.model_cfg.get("DIR_LIMIT_OFFSET", 0.0),
                                                  self.num_dir_bins)
        data_dict.update(batch_cls_preds=cls, batch_box_preds=boxes, cls_preds_normalized=False)

    def forward(self, data_dict):
        x = data_dict["st_features_2d"]
        if self.anchors_root is None:
            self.anchors_root = self._gen[0].generate_anchors(self._gen[1], device=x.device)[0]
        mask = self.get_anchor_mask(data_dict["points"], x.shape)
        self.anchors = [a[:, mask, ...] for a in self.anchors_root]
        pick = lambda t: t.permute(0, 2, 3, 1).contiguous()[:, mask, :]
        with torch.set_grad_enabled(self.training and torch.is_grad_enabled()):
            cls_preds, box_preds = pick(self.conv_cls(x)), pick(self.conv_box(x))
            dir_preds = pick(self.conv_dir_cls(x)) if self.conv_dir_cls is not None else None
        self.forward_ret_dict.update(cls_preds=cls_preds, box_preds=box_preds, dir_cls_preds=dir_preds)
        if self.training:
            self._train_targets(data_dict)
        if not self.training or self.predict_boxes_when_training:
            self._predict(data_dict, cls_preds, box_preds, dir_preds)
        return data_dict


def get_layer(dim, out_dim, init=None):
    """anchor_head_single.py:9-29: Conv3x3(dim, dim) + BatchNorm2d + ReLU + Conv1x1(dim, out_dim), reference init."""
    from .models import Conv2d
    conv = Conv2d(dim, dim, kernel_size=3, padding=1, bias=True)
    torch.nn.init.normal_(conv.weight, mean=0, std=0.001)
    conv2 = Conv2d(dim, out_dim, kernel_size=1, bias=True)
    if init is None:
        torch.nn.init.normal_(conv2.weight, mean=0, std=0.001)
    else:
        conv2.bias.data.fill_(init)
    return torch.nn.Sequential(conv, torch.nn.BatchNorm2d(dim), torch.nn.ReLU(), conv2)


class AnchorHeadSingleV2(AnchorHeadSingle):
    """AnchorHeadSingleV2 (cpd/models/dense_heads/anchor_head_single.py:31-192), the dense head both shipped dbscan / oyster
    configs select (tools/cfgs/models/waymo_unsupervised/voxel_rcnn_{dbscan,oyster}_single_train.yaml:37): a shared 3x3
    conv (input -> 64) + BN + ReLU, five `get_layer` branches on it (cls / reg / height / dim / ang), the 1x1 direction
    classifier on the input map, occupancy-masked anchors, generate_predicted_boxes. Same parameters and state_dict names as
    the reference (shared_conv.0/1, conv_cls.0/1/3, conv_reg..., conv_dir_cls).

    Eval forward = four C-ABI launches on channels-last rows, BatchNorm folded: the shared conv, the five branches' 3x3 convs
    fused along their output columns (64 -> 320), their five 1x1 convs as ONE block-diagonal 320 -> (A*nc + 7A) GEMM whose
    column order is the reference's torch.cat([reg, height, dim, ang]) order, and the direction classifier. Training mode runs
    the modules one by one (differentiable, cpd_amd/autograd_ops.py)."""

    SHARD_C = 64

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, conv_math="f16x2",
                 predict_boxes_when_training=True, **kwargs):
        torch.nn.Module.__init__(self)
        from .models import Conv2d
        self.model_cfg, self.num_class, self.class_names = model_cfg, num_class, list(class_names)
        self.predict_boxes_when_training = bool(predict_boxes_when_training)   # reference ctor arg (anchor_head_single.py:32-33)
        self.anchors = None
        self.range = [float(v) for v in point_cloud_range]
        self.voxel_size = (self.range[3] - self.range[0]) / float(grid_size[0])
        agc = model_cfg["ANCHOR_GENERATOR_CONFIG"]
        fms = [[int(grid_size[0]) // c["feature_map_stride"], int(grid_size[1]) // c["feature_map_stride"]] for c in agc]
        self._gen = (AnchorGenerator(self.range, agc), fms)
        self.anchors_root = None
        a = sum(len(c["anchor_rotations"]) * len(c["anchor_sizes"]) * len(c["anchor_bottom_heights"]) for c in agc)
        self.num_anchors_per_location = a
        c = self.SHARD_C
        self.shared_conv = torch.nn.Sequential(Conv2d(input_channels, c, kernel_size=3, padding=1, bias=True),
                                               torch.nn.BatchNorm2d(c), torch.nn.ReLU(inplace=True))
        self.conv_cls = get_layer(c, a * num_class, -4.59)
        self.conv_reg = get_layer(c, a * 2)
        self.conv_height = get_layer(c, a * 1)
        self.conv_dim = get_layer(c, a * 3)
        self.conv_ang = get_layer(c, a * 1)
        self.num_dir_bins = model_cfg.get("NUM_DIR_BINS", 2)
        self.conv_dir_cls = Conv2d(input_channels, a * self.num_dir_bins, kernel_size=1) \
            if model_cfg.get("USE_DIRECTION_CLASSIFIER", None) is not None else None
        self.conv_math = conv_math
        self.forward_ret_dict = {}
        self._fused = None

    BRANCHES = ("conv_cls", "conv_reg", "conv_height", "conv_dim", "conv_ang")

    def _fuse(self, dev):
        """Folded / fused weight images of the eval path (rebuilt when a parameter changes)."""
        from .engine import _fold_bn
        params = list(self.parameters()) + list(self.buffers())
        key = tuple((p._version, p.data_ptr()) for p in params)
        if self._fused is not None and self._fused[0] == key:
            return self._fused[1]
        sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
        c = self.SHARD_C

        def kio(w):                                                  # (Cout, Cin, k, k) -> [k*k, Cin, Cout]
            return w.permute(2, 3, 1, 0).reshape(w.shape[2] * w.shape[3], w.shape[1], w.shape[0]).contiguous()

        f = {}
        s, t = _fold_bn(sd, "shared_conv.1", 1e-5, sd["shared_conv.0.bias"])
        f["shared"] = (ops.pack_weight(kio(sd["shared_conv.0.weight"]).to(dev)), s.to(dev), t.to(dev), sd["shared_conv.0.weight"].shape[1], c)
        w1, s1, t1 = [], [], []
        outs = [sd[b + ".3.weight"].shape[0] for b in self.BRANCHES]
        w2 = torch.zeros(1, c * len(self.BRANCHES), sum(outs))
        b2 = torch.zeros(sum(outs))
        col = 0
        for i, b in enumerate(self.BRANCHES):
            w1.append(kio(sd[b + ".0.weight"]))
            s, t = _fold_bn(sd, b + ".1", 1e-5, sd[b + ".0.bias"])
            s1.append(s); t1.append(t)
            w2[0, i * c:(i + 1) * c, col:col + outs[i]] = sd[b + ".3.weight"].reshape(outs[i], c).t()
            b2[col:col + outs[i]] = sd[b + ".3.bias"]
            col += outs[i]
        f["first"] = (ops.pack_weight(torch.cat(w1, dim=2).to(dev)), torch.cat(s1).to(dev), torch.cat(t1).to(dev), c, c * len(self.BRANCHES))
        f["second"] = (ops.pack_weight(w2.to(dev)), None, b2.to(dev), c * len(self.BRANCHES), sum(outs))
        f["n_cls"] = outs[0]
        if self.conv_dir_cls is not None:
            wd = sd["conv_dir_cls.weight"]
            f["dir"] = (ops.pack_weight(kio(wd).to(dev)), None, sd["conv_dir_cls.bias"].to(dev), wd.shape[1], wd.shape[0])
        self._fused = (key, f)
        return f

    def _heads_eval(self, x):
        """(cls rows, box rows, dir rows | None), each [B*H*W, channels], through the fused launches."""
        from .models import _rows
        b, cin, h, w = x.shape
        dev = x.device
        f = self._fuse(dev)
        rows = _rows(x.float())
        n = b * h * w
        nbr, _, _ = ops.rulebook_conv2d(b, h, w, 3, 3, 1, 1, dev)
        m = self.conv_math

        def run(name, inp, table, kv, relu):
            pw, s, t, ci, co = f[name]
            return ops.gather_conv(inp, ci, pw, table, kv, n, co, s, t, None, relu, dense=True, math=m)

        shared = run("shared", rows, nbr, 9, True)
        first = run("first", shared, nbr, 9, True)
        second = run("second", first, None, 1, False)
        dirs = run("dir", rows, None, 1, False) if "dir" in f else None
        return second[:, :f["n_cls"]], second[:, f["n_cls"]:], dirs

    def forward(self, data_dict):
        x = data_dict["st_features_2d"]
        b, _, h, w = x.shape
        if self.anchors_root is None:
            self.anchors_root = self._gen[0].generate_anchors(self._gen[1], device=x.device)[0]
        mask = self.get_anchor_mask(data_dict["points"], x.shape)
        self.anchors = [a[:, mask, ...] for a in self.anchors_root]
        if self.training:
            shard = self.shared_conv(x)
            pick = lambda t: t.permute(0, 2, 3, 1).contiguous()[:, mask, :]
            cls_preds = pick(self.conv_cls(shard))
            box_preds = pick(torch.cat([self.conv_reg(shard), self.conv_height(shard), self.conv_dim(shard), self.conv_ang(shard)], dim=1))
            dir_preds = pick(self.conv_dir_cls(x)) if self.conv_dir_cls is not None else None
        else:
            with torch.no_grad():
                cls_r, box_r, dir_r = self._heads_eval(x)
                pick = lambda r: r.reshape(b, h, w, r.shape[1])[:, mask, :]
                cls_preds, box_preds = pick(cls_r), pick(box_r)
                dir_preds = pick(dir_r) if dir_r is not None else None
        self.forward_ret_dict.update(cls_preds=cls_preds, box_preds=box_preds, dir_cls_preds=dir_preds)
        if self.training:                                  # anchor_head_single.py:176-181: targets on the MASKED anchors
            self._train_targets(data_dict)
        if not self.training or self.predict_boxes_when_training:
            self._predict(data_dict, cls_preds, box_preds, dir_preds)
        return data_dict
