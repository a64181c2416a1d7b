"""Training branch of the second stage (SURVEY 8f-2's consumer): `VoxelRCNNProtoHead` with its proposal-target sampling,
canonical transformation, RoI losses and the prototype ("MM") pooling branch.

  ProposalTargetLayer            cpd/models/roi_heads/target_assigner/proposal_target_layer.py:7-427
  assign_targets                 cpd/models/roi_heads/roi_head_template.py:116-146
  VoxelRCNNProtoHead             cpd/models/roi_heads/voxel_rcnn_head.py:16-662 (forward, roi_grid_pool(_mm), get_loss, proto_loss,
                                 get_box_reg_layer_loss, get_box_cls_layer_loss)
  bb_loss                        cpd/utils/bbloss.py:4-48;  get_corner_loss_lidar cpd/utils/loss_utils.py:210-233

Where the work is: the RoI x GT 3-D IoU matrices (`cpd_boxes_iou3d`), proposal NMS (`cpd_nms_batch`), the neighbour queries and
the differentiable grouping (`cpd_voxel_query*`, `cpd_group_points`, `cpd_group_points_grad`) are C-ABI kernels; the FC stacks,
BatchNorm with batch statistics and the loss formulas are torch ops on the device, as they are in the reference. The branches
built are the ones the shipped config selects (voxel_rcnn_cproto_center.yaml: CLS_SCORE_TYPE roi_iou or cls, scalar thresholds,
BinaryCrossEntropy, smooth-l1, corner regularisation); the per-class threshold variants (roi_iou_x / roi_ioud*) raise.
The sampler draws its random numbers with the same calls in the same order as the reference (np.random.permutation / rand,
torch.randint on the CPU generator), so equal seeds give equal samples.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .autograd_ops import HipBatchNorm1d, HipConv1d, HipLinear
from . import roi_pool as rp

TWO_PI = 2 * np.pi


def _cfg(c, key, default=None):
    if isinstance(c, dict):
        return c.get(key, default)
    return getattr(c, key, default)


# ------------------------------------------------------------------------------------------------ small geometry helpers
def rotate_points_along_z(points, angle):
    """common_utils.rotate_points_along_z (common_utils.py:35-57): points (B, N, 3+C), angle (B)."""
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = torch.zeros_like(cosa), torch.ones_like(cosa)
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    return torch.cat((torch.matmul(points[:, :, 0:3], rot), points[:, :, 3:]), dim=-1)


def boxes_to_corners_3d(boxes):
    """box_utils.boxes_to_corners_3d (box_utils.py:27-52): (N, 7) -> (N, 8, 3)."""
    template = boxes.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    corners = boxes[:, None, 3:6].repeat(1, 8, 1) * template[None]
    corners = rotate_points_along_z(corners.view(-1, 8, 3), boxes[:, 6]).view(-1, 8, 3)
    return corners + boxes[:, None, 0:3]


def residual_encode(boxes, anchors):
    """ResidualCoder.encode_torch (box_coder_utils.py:13-44), code_size 7, out of place."""
    a = torch.cat([anchors[:, 0:3], anchors[:, 3:6].clamp_min(1e-5), anchors[:, 6:7]], dim=-1)
    g = torch.cat([boxes[:, 0:3], boxes[:, 3:6].clamp_min(1e-5), boxes[:, 6:7]], dim=-1)
    diag = torch.sqrt(a[:, 3] ** 2 + a[:, 4] ** 2)
    return torch.stack([(g[:, 0] - a[:, 0]) / diag, (g[:, 1] - a[:, 1]) / diag, (g[:, 2] - a[:, 2]) / a[:, 5], torch.log(g[:, 3] / a[:, 3]),
                        torch.log(g[:, 4] / a[:, 4]), torch.log(g[:, 5] / a[:, 5]), g[:, 6] - a[:, 6]], dim=-1)


def residual_decode(enc, anchors):
    """ResidualCoder.decode_torch (box_coder_utils.py:46-78), code_size 7; differentiable in `enc`."""
    xa, ya, za, dxa, dya, dza, ra = torch.unbind(anchors, dim=-1)
    xt, yt, zt, dxt, dyt, dzt, rt = torch.unbind(enc, dim=-1)
    diag = torch.sqrt(dxa ** 2 + dya ** 2)
    return torch.stack([xt * diag + xa, yt * diag + ya, zt * dza + za, torch.exp(dxt) * dxa, torch.exp(dyt) * dya, torch.exp(dzt) * dza,
                        rt + ra], dim=-1)


def smooth_l1(diff, beta):
    n = diff.abs()
    return n if beta < 1e-5 else torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)


def corner_loss_lidar(pred, gt):
    """loss_utils.get_corner_loss_lidar (loss_utils.py:210-233): (N, 7), (N, 7) -> (N)."""
    pc, gc = boxes_to_corners_3d(pred), boxes_to_corners_3d(gt)
    flip = gt.clone()
    flip[:, 6] += np.pi
    fc = boxes_to_corners_3d(flip)
    dist = torch.min(torch.norm(pc - gc, dim=2), torch.norm(pc - fc, dim=2))
    return smooth_l1(dist, 1.0).mean(dim=1)


def _limit(ang):                                                     # bbloss.limit (bbloss.py:4-11)
    ang = ang % TWO_PI
    ang = torch.where(ang > np.pi, ang - TWO_PI, ang)
    return torch.where(ang < -np.pi, ang + TWO_PI, ang)


def _axis_iou(x, w, y, l):                                           # bbloss.compute_iou (bbloss.py:20-28)
    hi1, lo1, hi2, lo2 = x + w * 0.5, x - w * 0.5, y + l * 0.5, y - l * 0.5
    inter = (torch.min(hi1, hi2) - torch.max(lo1, lo2)).clamp_min(0.)
    span = (torch.max(hi1, hi2) - torch.min(lo1, lo2)).clamp_min(0.)
    return inter / span


def bb_loss(pred, target):
    """bbloss.bb_loss (bbloss.py:30-48): per-axis 1-D IoUs x heading agreement + heading and centre penalties, x 1.5."""
    iou = _axis_iou(pred[..., 0], pred[..., 3], target[..., 0], target[..., 3]) * _axis_iou(pred[..., 1], pred[..., 4], target[..., 1], target[..., 4]) \
        * _axis_iou(pred[..., 2], pred[..., 5], target[..., 2], target[..., 5])
    iou = iou * (1 - torch.abs(torch.sin(_limit(pred[..., 6]) - _limit(target[..., 6]))))
    angle_factor = 1.25 * (1.0 - torch.abs(torch.cos(pred[:, -1] - target[:, -1])))
    dist2 = torch.pow(target[:, 0:3] - pred[:, 0:3], 2).sum(-1)
    return (1 - iou + angle_factor + dist2) * 1.5


# ------------------------------------------------------------------------------------------------ proposal target layer
class ProposalTargetLayer(nn.Module):
    def __init__(self, roi_sampler_cfg):
        super().__init__()
        self.cfg = roi_sampler_cfg

    def g(self, key, default=None):
        return _cfg(self.cfg, key, default)

    @staticmethod
    def get_max_iou_with_same_class(rois, roi_labels, gt_boxes, gt_labels):
        """l.395-427: 3-D IoU of every RoI with the boxes of its own class only."""
        max_overlaps = rois.new_zeros(rois.shape[0])
        gt_assignment = roi_labels.new_zeros(roi_labels.shape[0])
        for k in range(int(gt_labels.min()), int(gt_labels.max()) + 1):
            rm, gm = roi_labels == k, gt_labels == k
            if int(rm.sum()) > 0 and int(gm.sum()) > 0:
                origin = gm.nonzero().view(-1)
                iou = ops.boxes_iou3d(rois[rm][:, 0:7].contiguous(), gt_boxes[gm][:, 0:7].contiguous())
                best, arg = torch.max(iou, dim=1)
                max_overlaps[rm] = best
                gt_assignment[rm] = origin[arg]
        return max_overlaps, gt_assignment

    @staticmethod
    def sample_bg_inds(hard, easy, num, hard_ratio):
        """l.364-393; torch.randint draws on the CPU generator, hard negatives first."""
        def draw(pool, k):
            return pool[torch.randint(low=0, high=pool.numel(), size=(k,)).long().to(pool.device)]
        if hard.numel() > 0 and easy.numel() > 0:
            n_hard = min(int(num * hard_ratio), len(hard))
            h = draw(hard, n_hard)
            return torch.cat([h, draw(easy, num - n_hard)], dim=0)
        if hard.numel() > 0:
            return draw(hard, num)
        if easy.numel() > 0:
            return draw(easy, num)
        raise NotImplementedError

    def subsample_rois(self, max_overlaps, gts=None):
        """l.293-362: scalar thresholds, or (`gts` = every RoI's assigned ground-truth row, CLS_SCORE_TYPE roi_iou_x / roi_ioud_x) the
        per-class threshold LISTS, class c + 1 of the assigned box selecting entry c (l.300-309, 318-325)."""
        per_image = self.g("ROI_PER_IMAGE")
        fg_per_image = int(np.round(self.g("FG_RATIO") * per_image))
        lo = self.g("CLS_BG_THRESH_LO")
        easy = (max_overlaps < lo).nonzero().view(-1)
        if gts is None:
            fg_thresh = min(self.g("REG_FG_THRESH"), self.g("CLS_FG_THRESH"))
            fg = (max_overlaps >= fg_thresh).nonzero().view(-1)
            hard = ((max_overlaps < self.g("REG_FG_THRESH")) & (max_overlaps >= lo)).nonzero().view(-1)
        else:
            reg_t, cls_t = list(self.g("REG_FG_THRESH")), list(self.g("CLS_FG_THRESH"))
            fg_m = torch.zeros_like(max_overlaps, dtype=torch.bool)
            hard_m = torch.zeros_like(max_overlaps, dtype=torch.bool)
            for c in range(len(cls_t)):
                own = gts[..., -1] == (c + 1)
                fg_m |= (max_overlaps >= min(reg_t[c], cls_t[c])) & own
            for c in range(len(reg_t)):
                own = gts[..., -1] == (c + 1)
                hard_m |= (max_overlaps < reg_t[c]) & (max_overlaps >= lo) & own
            fg, hard = fg_m.nonzero().view(-1), hard_m.nonzero().view(-1)
        n_fg, n_bg = fg.numel(), hard.numel() + easy.numel()
        if n_fg > 0 and n_bg > 0:
            k = min(fg_per_image, n_fg)
            perm = torch.from_numpy(np.random.permutation(n_fg)).long().to(fg.device)
            fg = fg[perm[:k]]
            bg = self.sample_bg_inds(hard, easy, per_image - k, self.g("HARD_BG_RATIO"))
        elif n_fg > 0:
            pick = torch.from_numpy(np.floor(np.random.rand(per_image) * n_fg)).long().to(fg.device)
            fg = fg[pick]
            bg = fg.new_zeros((0,))
        elif n_bg > 0:
            bg = self.sample_bg_inds(hard, easy, per_image, self.g("HARD_BG_RATIO"))
        else:
            raise NotImplementedError("no foreground and no background RoIs (max overlap %f..%f)" % (float(max_overlaps.min()),
                                                                                                      float(max_overlaps.max())))
        return torch.cat((fg, bg), dim=0)

    def sample_rois_for_rcnn(self, batch_dict, ind=""):
        """l.198-291."""
        b = batch_dict["batch_size"]
        rois, scores, labels, gt_boxes = batch_dict["rois"], batch_dict["roi_scores"], batch_dict["roi_labels"], batch_dict["gt_boxes" + ind]
        per_image = self.g("ROI_PER_IMAGE")
        extra_keys = [k for k in ("css_score", "outline_cls_mask", "outline_reg_mask") if k in batch_dict]
        extra = {k: rois.new_zeros(b, per_image) for k in extra_keys}
        out_rois = rois.new_zeros(b, per_image, rois.shape[-1])
        out_gt = rois.new_zeros(b, per_image, gt_boxes.shape[-1])
        out_iou, out_scores = rois.new_zeros(b, per_image), rois.new_zeros(b, per_image)
        out_labels = rois.new_zeros((b, per_image), dtype=torch.long)
        for i in range(b):
            cur_gt = gt_boxes[i]
            nz = (cur_gt.sum(1) != 0).nonzero()
            k = int(nz[-1]) if nz.numel() else 0                    # trailing all-zero rows are padding; at least one row stays
            cur_gt = cur_gt[:k + 1]
            if self.g("SAMPLE_ROI_BY_EACH_CLASS", False):
                overlaps, assign = self.get_max_iou_with_same_class(rois[i], labels[i], cur_gt[:, 0:7], cur_gt[:, -1].long())
            else:
                overlaps, assign = torch.max(ops.boxes_iou3d(rois[i][:, 0:7].contiguous(), cur_gt[:, 0:7].contiguous()), dim=1)
            by_class = self.g("CLS_SCORE_TYPE") in ("roi_iou_x", "roi_ioud_x")            # l.258-261
            sel = self.subsample_rois(overlaps, gts=cur_gt[assign] if by_class else None)
            out_rois[i], out_labels[i], out_iou[i], out_scores[i] = rois[i][sel], labels[i][sel], overlaps[sel], scores[i][sel]
            out_gt[i] = cur_gt[assign[sel]]
            for key in extra_keys:
                extra[key][i] = batch_dict[key][i][:k + 1][assign[sel]]
        return out_rois, out_gt, out_iou, out_scores, out_labels, extra

    @torch.no_grad()
    def forward(self, batch_dict, ind=""):
        """l.32-196: CLS_SCORE_TYPE cls / roi_iou / roi_ioud (scalar thresholds) and roi_iou_x / roi_ioud_x (per-class threshold lists
        picked by the class of the RoI's assigned ground truth, l.57-80, 128-184; ENABLE_HARD_SAMPLING draws np.random.randint once per
        class, after the sampling's own draws, and -- as the reference does -- switches whole BATCH rows on: `mask_prob[ints]` indexes
        dimension 0 of the (B, ROI_PER_IMAGE) mask)."""
        rois, gt_of_rois, ious, scores, labels, extra = self.sample_rois_for_rcnn(batch_dict, ind)
        kind = self.g("CLS_SCORE_TYPE")

        def direction_weight():
            a, g_ = _limit(rois[..., 6]), _limit(gt_of_rois[..., 6])
            d = torch.abs(a - g_)
            w = 1 - torch.min(d, TWO_PI - d) / np.pi
            lo, hi = self.g("DIRECTION_MIN"), self.g("DIRECTION_MAX")
            return (torch.clamp(w, lo, hi) - lo) / (hi - lo)

        if kind in ("roi_iou_x", "roi_ioud_x"):
            gt_cls = gt_of_rois[..., -1]
            reg_valid = torch.zeros_like(ious, dtype=torch.long)
            reg_t = list(self.g("REG_FG_THRESH"))
            for c in range(len(reg_t)):
                own = gt_cls == (c + 1)
                this = ((ious > reg_t[c]) & own).long()
                if self.g("ENABLE_HARD_SAMPLING", False):
                    hard = (ious < reg_t[c]) & (ious > self.g("HARD_SAMPLING_THRESH")[c]) & own
                    every = int(1 / self.g("HARD_SAMPLING_RATIO")[c])
                    on = torch.zeros_like(hard)
                    on[list(range(np.random.randint(0, every), on.shape[0], every))] = True
                    this = this + (hard & on).long()
                reg_valid += this
            fg_l, bg_l = list(self.g("CLS_FG_THRESH")), list(self.g("CLS_BG_THRESH"))
            cls_labels = torch.zeros_like(ious)
            dw = direction_weight() if kind == "roi_ioud_x" else None
            for c in range(len(bg_l)):
                fg, bg = ious > fg_l[c], ious < bg_l[c]
                mid = (~fg) & (~bg)
                lab = fg.float()
                lab[mid] = (ious[mid] - bg_l[c]) / (fg_l[c] - bg_l[c])
                if dw is not None:
                    lab = lab * dw
                own = gt_cls == (c + 1)
                cls_labels[own] = lab[own]
            return {"rois": rois, "gt_of_rois": gt_of_rois, "gt_iou_of_rois": ious, "roi_scores": scores, "roi_labels": labels,
                    "reg_valid_mask": reg_valid, "rcnn_cls_labels": cls_labels, "additional_data": extra}
        reg_valid = (ious > self.g("REG_FG_THRESH")).long()
        fg_t, bg_t = self.g("CLS_FG_THRESH"), self.g("CLS_BG_THRESH")
        if kind == "cls":
            cls_labels = (ious > fg_t).long()
            cls_labels[(ious > bg_t) & (ious < fg_t)] = -1
        elif kind in ("roi_iou", "roi_ioud"):
            fg, bg = ious > fg_t, ious < bg_t
            mid = (~fg) & (~bg)
            cls_labels = fg.float()
            cls_labels[mid] = (ious[mid] - bg_t) / (fg_t - bg_t)
            if kind == "roi_ioud":
                cls_labels = cls_labels * direction_weight()
        else:
            raise NotImplementedError(kind)
        return {"rois": rois, "gt_of_rois": gt_of_rois, "gt_iou_of_rois": ious, "roi_scores": scores, "roi_labels": labels,
                "reg_valid_mask": reg_valid, "rcnn_cls_labels": cls_labels, "additional_data": extra}


def assign_targets(target_layer, batch_dict, ind=""):
    """RoIHeadTemplate.assign_targets (roi_head_template.py:116-146): sampling, then the ground truth of every RoI expressed in
    the RoI's canonical frame (centre at the origin, heading zero, flipped when it points the other way)."""
    b = batch_dict["batch_size"]
    with torch.no_grad():
        t = target_layer(batch_dict, ind)
    rois, gt = t["rois"], t["gt_of_rois"]
    t["gt_of_rois_src"] = gt.clone().detach()
    ry = rois[:, :, 6] % TWO_PI
    gt = gt.clone()
    gt[:, :, 0:3] = gt[:, :, 0:3] - rois[:, :, 0:3]
    gt[:, :, 6] = gt[:, :, 6] - ry
    gt = rotate_points_along_z(gt.view(-1, 1, gt.shape[-1]), -ry.view(-1)).view(b, -1, gt.shape[-1])
    h = gt[:, :, 6] % TWO_PI
    opposite = (h > np.pi * 0.5) & (h < np.pi * 1.5)
    h = torch.where(opposite, (h + np.pi) % TWO_PI, h)
    h = torch.where(h > np.pi, h - TWO_PI, h)
    gt[:, :, 6] = torch.clamp(h, min=-np.pi / 2, max=np.pi / 2)
    t["gt_of_rois"] = gt
    return t


# ------------------------------------------------------------------------------------------------ the head
def _fc_stack(pre, widths, dp, final=None, dropout_between=True):
    layers = []
    for k, w in enumerate(widths):
        layers += [HipLinear(pre, w, bias=False), HipBatchNorm1d(w), nn.ReLU()]
        pre = w
        if dropout_between and k != len(widths) - 1 and dp > 0:
            layers.append(nn.Dropout(dp))
    if final is not None:
        layers.append(HipLinear(pre, final, bias=True))
    return nn.Sequential(*layers)


class VoxelRCNNProtoHead(nn.Module):
    """Same constructor arguments, parameter names and batch_dict contract as the reference class. `forward` in training mode
    fills `forward_ret_dict['targets_dict0' / 'targets_dict1']`; `get_loss()` returns (rcnn_loss, tb_dict)."""

    def __init__(self, input_channels, model_cfg, point_cloud_range=None, voxel_size=None, num_frames=1, num_class=1, **kwargs):
        super().__init__()
        self.model_cfg, self.point_cloud_range, self.voxel_size, self.num_class = model_cfg, point_cloud_range, voxel_size, num_class
        self.pool_cfg, self.pool_cfg_mm = model_cfg["ROI_GRID_POOL"], model_cfg["ROI_GRID_POOL_PROTO"]
        self.code_size = 7

        def pool_layers(cfg, feat_mult):
            layers, c_out = nn.ModuleList(), 0
            for src in cfg["FEATURES_SOURCE"]:
                lc = cfg["POOL_LAYERS"][src]
                mlps = [[input_channels[src] * feat_mult] + list(m) for m in lc["MLPS"]]
                layers.append(rp.NeighborVoxelSAModuleMSG(query_ranges=lc["QUERY_RANGES"], nsamples=lc["NSAMPLE"], radii=lc["POOL_RADIUS"],
                                                          mlps=mlps, pool_method=lc["POOL_METHOD"]))
                c_out += sum(m[-1] for m in mlps)
            return layers, c_out

        self.roi_grid_pool_layers, c_out = pool_layers(self.pool_cfg, 1)
        self.roi_grid_pool_layers_mm, c_out_mm = pool_layers(self.pool_cfg_mm, _cfg(self.pool_cfg_mm, "FEAT_NUM", 1))
        dp, sfc = model_cfg["DP_RATIO"], list(model_cfg["SHARED_FC"])
        self.shared_fc_layers = _fc_stack(self.pool_cfg["GRID_SIZE"] ** 3 * c_out, sfc, dp)
        self.shared_fc_layers_mm = _fc_stack(self.pool_cfg_mm["GRID_SIZE"] ** 3 * c_out_mm, sfc, dp)
        self.cls_layers = _fc_stack(sfc[-1], list(model_cfg["CLS_FC"]), dp, num_class)
        self.reg_layers = _fc_stack(sfc[-1], list(model_cfg["REG_FC"]), dp, self.code_size * num_class)
        self.cls_layers_P = _fc_stack(sfc[-1], list(model_cfg["CLS_FC"]), dp, num_class)
        self.reg_layers_P = _fc_stack(sfc[-1], list(model_cfg["REG_FC"]), dp, self.code_size * num_class)
        self.proposal_target_layer = ProposalTargetLayer(model_cfg["TARGET_CONFIG"])
        cw = model_cfg["LOSS_CONFIG"]["LOSS_WEIGHTS"]["code_weights"]
        self.register_buffer("code_weights", torch.tensor(cw, dtype=torch.float32), persistent=False)
        self.forward_ret_dict = {}
        self.iter = 1
        self.init_weights()

    def init_weights(self):
        """voxel_rcnn_head.py:168-184: xavier on the detection branch's stacks, N(0, 0.01) on its two output layers; the
        prototype branch (`*_P`, `*_mm`) keeps torch's default initialisation, as in the reference."""
        for stack in (self.cls_layers, self.reg_layers, self.shared_fc_layers):
            for m in stack.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_normal_(m.weight)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
        for stack in (self.cls_layers, self.reg_layers):
            nn.init.normal_(stack[-1].weight, 0, 0.01)
            nn.init.constant_(stack[-1].bias, 0)

    # ---- pooling -------------------------------------------------------------------------------------------------------
    def _pool(self, batch_dict, key, cfg, layers):
        levels = {}
        for name in cfg["FEATURES_SOURCE"]:
            t = batch_dict[key][name]
            levels[name] = t if isinstance(t, tuple) else (t.features, t.indices, list(t.spatial_shape))
        return rp.roi_grid_pool(batch_dict["rois"], levels, batch_dict["multi_scale_3d_strides"], dict(zip(cfg["FEATURES_SOURCE"], layers)),
                                cfg["GRID_SIZE"], self.voxel_size, self.point_cloud_range, batch_dict["batch_size"])

    def roi_grid_pool(self, batch_dict):
        return self._pool(batch_dict, "multi_scale_3d_features", self.pool_cfg, self.roi_grid_pool_layers)

    def roi_grid_pool_mm(self, batch_dict):
        return self._pool(batch_dict, "multi_scale_3d_features_mm", self.pool_cfg_mm, self.roi_grid_pool_layers_mm)

    def proposal_layer(self, batch_dict, nms_config):
        if batch_dict.get("rois", None) is not None:
            return batch_dict
        rois, scores, labels, _ = rp.proposal_layer(batch_dict["batch_box_preds"], batch_dict["batch_cls_preds"], nms_config["NMS_THRESH"],
                                                    nms_config["NMS_PRE_MAXSIZE"], nms_config["NMS_POST_MAXSIZE"], first_rows="auto", device_fallback=True)
        batch_dict.update(rois=rois, roi_scores=scores, roi_labels=labels, has_class_labels=batch_dict["batch_cls_preds"].shape[-1] > 1)
        return batch_dict

    # ---- forward -------------------------------------------------------------------------------------------------------
    def forward(self, batch_dict):
        """voxel_rcnn_head.py:581-662."""
        self.proposal_layer(batch_dict, self.model_cfg["NMS_CONFIG"]["TRAIN" if self.training else "TEST"])
        if self.training:
            targets = assign_targets(self.proposal_target_layer, batch_dict)
            batch_dict["rois"], batch_dict["roi_labels"] = targets["rois"], targets["roi_labels"]

            def copy(t):
                return {k: ({kk: vv.clone() for kk, vv in v.items()} if k == "additional_data" else v.clone()) for k, v in t.items()}
            t0, t1 = copy(targets), copy(targets)
        pooled = self.roi_grid_pool(batch_dict)
        shared = self.shared_fc_layers(pooled.reshape(pooled.shape[0], -1))
        rcnn_cls, rcnn_reg = self.cls_layers(shared), self.reg_layers(shared)
        if not self.training:
            cls, boxes = rp.VoxelRCNNHead.generate_predicted_boxes(batch_dict["batch_size"], batch_dict["rois"], rcnn_cls, rcnn_reg)
            batch_dict.update(batch_box_preds=boxes, batch_cls_preds=cls, cls_preds_normalized=False)
            return batch_dict
        t0.update(rcnn_cls=rcnn_cls, rcnn_reg=rcnn_reg, shared_features=shared)
        pooled_p = self.roi_grid_pool_mm(batch_dict)
        shared_p = self.shared_fc_layers_mm(pooled_p.reshape(pooled_p.shape[0], -1))
        t1.update(rcnn_cls=self.cls_layers_P(shared_p), rcnn_reg=self.reg_layers_P(shared_p), shared_features=shared_p)
        self.forward_ret_dict["targets_dict0"], self.forward_ret_dict["targets_dict1"] = t0, t1
        return batch_dict

    # ---- losses --------------------------------------------------------------------------------------------------------
    def get_box_cls_layer_loss(self, t):
        """l.529-554 (BinaryCrossEntropy; the prototype confidence `css_score` weights the numerator only)."""
        lc = self.model_cfg["LOSS_CONFIG"]
        if lc["CLS_LOSS"] != "BinaryCrossEntropy":
            raise NotImplementedError(lc["CLS_LOSS"])
        labels = t["rcnn_cls_labels"].view(-1)
        valid = (labels >= 0).float()
        bce = F.binary_cross_entropy(torch.sigmoid(t["rcnn_cls"].view(-1)), labels.float(), reduction="none")
        loss = (bce * valid * t["additional_data"]["css_score"].view(-1)).sum() / torch.clamp(valid.sum(), min=1.0)
        loss = loss * lc["LOSS_WEIGHTS"]["rcnn_cls_weight"]
        return loss, {"rcnn_loss_cls": loss.item()}

    def get_box_reg_layer_loss(self, t):
        """l.461-527."""
        lc, cs = self.model_cfg["LOSS_CONFIG"], self.code_size
        if lc["REG_LOSS"] != "smooth-l1":
            raise NotImplementedError(lc["REG_LOSS"])
        gt_ct = t["gt_of_rois"][..., 0:cs].reshape(-1, cs)
        gt_src = t["gt_of_rois_src"][..., 0:cs].reshape(-1, cs)
        reg, rois = t["rcnn_reg"].view(gt_ct.shape[0], -1), t["rois"].reshape(-1, cs)
        fg = (t["reg_valid_mask"].view(-1) * t["additional_data"]["css_score"].view(-1)) > 0
        fg_sum = int(fg.long().sum())
        anchor = rois.clone().detach()
        anchor[:, 0:3] = 0
        anchor[:, 6] = 0
        target = residual_encode(gt_ct, anchor)
        target = torch.where(torch.isnan(target), reg, target)
        loss = smooth_l1((reg - target) * self.code_weights.view(1, -1), 1.0 / 9.0)
        loss = (loss * fg.unsqueeze(-1).float()).sum() / max(fg_sum, 1) * lc["LOSS_WEIGHTS"]["rcnn_reg_weight"]
        tb = {"rcnn_loss_reg": loss.item()}
        if lc["CORNER_LOSS_REGULARIZATION"] and fg_sum > 0:
            fg_rois = rois[fg]
            anchors = fg_rois.clone().detach()
            anchors[:, 0:3] = 0
            boxes = residual_decode(reg[fg], anchors)
            boxes = rotate_points_along_z(boxes.unsqueeze(1), fg_rois[:, 6]).squeeze(1)
            boxes = torch.cat([boxes[:, 0:3] + fg_rois[:, 0:3], boxes[:, 3:]], dim=-1)
            corner = corner_loss_lidar(boxes[:, 0:7], gt_src[fg][:, 0:7]).mean() * lc["LOSS_WEIGHTS"]["rcnn_corner_weight"]
            loss = loss + corner
            tb["rcnn_loss_corner"] = corner.item()
        return loss, tb

    def proto_loss(self, t0, t1):
        """l.388-459: the detection branch is pulled towards its canonical ground truth and towards the (detached) prototype
        branch, in box space (bb_loss) and in feature space (cosine similarity), ramped up over the first 5000 iterations."""
        cs = self.code_size
        fg = t0["reg_valid_mask"].view(-1) > 0
        gt_ct = t0["gt_of_rois"].clone().view(-1, t0["gt_of_rois"].shape[-1])[:, 0:7]
        css = t0["additional_data"]["css_score"].view(-1)

        def decoded(t):
            anchor = t["rois"].clone().view(-1, cs)[:, 0:7]
            anchor[:, 0:3] = 0
            anchor[:, 6] = 0
            return residual_decode(t["rcnn_reg"], anchor).view(-1, cs)
        p0, p1 = decoded(t0), decoded(t1).clone().detach()
        if int(fg.sum()) == 0:
            b0 = b1 = 0
        else:
            b0 = (bb_loss(p0[fg], gt_ct[fg]) * css[fg]).sum() / (fg.sum() + 1)
            b1 = (bb_loss(p0[fg], p1[fg]) * css[fg]).sum() / (fg.sum() + 1)
        begin, end, max_iter = 0.00001, 0.2, 5000
        self.iter = min(self.iter, max_iter)
        w = (self.iter / max_iter) * (end - begin) + begin
        self.iter += 1
        b1 = b1 * w
        sim = -torch.cosine_similarity(t0["shared_features"], t1["shared_features"].clone().detach(), dim=-1)
        valid = (t0["rcnn_cls_labels"].view(-1) >= 0).float() * css
        feat = (sim * valid).sum() / torch.clamp(valid.sum(), min=1.0) * w
        return b0 + b1 * w + feat

    def get_loss(self, tb_dict=None):
        """l.556-579."""
        tb_dict = {} if tb_dict is None else tb_dict
        t0, t1 = self.forward_ret_dict["targets_dict0"], self.forward_ret_dict["targets_dict1"]
        cls0, _ = self.get_box_cls_layer_loss(t0)
        reg0, _ = self.get_box_reg_layer_loss(t0)
        cls1, _ = self.get_box_cls_layer_loss(t1)
        reg1, _ = self.get_box_reg_layer_loss(t1)
        proto = 0.5 * cls1 + 0.5 * reg1 + self.proto_loss(t0, t1)
        loss = cls0 + reg0 + proto * self.model_cfg["LOSS_CONFIG"]["LOSS_WEIGHTS"]["rcnn_proto_weight"]
        tb_dict["rcnn_loss"] = loss.item()
        return loss, tb_dict
