"""CenterHead target assignment and loss (the torch side of the train step; SURVEY 8a-10 "loss via
torch"). Vectorised, device-resident restatement of
  CenterHead.assign_target_of_single_head / assign_targets  cpd/models/dense_heads/center_head.py:103-219
  centernet_utils.gaussian_radius / draw_gaussian_to_heatmap  cpd/models/model_utils/centernet_utils.py:9-69
  CenterHead.get_loss                                         center_head.py:221-250
  loss_utils.neg_loss_cornernet / _reg_loss                   cpd/utils/loss_utils.py:265-386
The reference assigns targets box by box on the CPU (center_head.py:204); here every box's gaussian
patch is rasterised at once and max-merged into the heat map with one scatter-amax."""
import torch


def gaussian_radius(height, width, min_overlap=0.5):
    a1 = 1
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + (b1 ** 2 - 4 * a1 * c1).sqrt()) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + (b2 ** 2 - 4 * a2 * c2).sqrt()) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + (b3 ** 2 - 4 * a3 * c3).sqrt()) / 2
    return torch.min(torch.min(r1, r2), r3)


def assign_targets(gt_boxes, feature_map_size, point_cloud_range, voxel_size, num_classes, feature_map_stride=8,
                   num_max_objs=500, gaussian_overlap=0.1, min_radius=2):
    """gt_boxes [B, M, 8 + E] (x,y,z,dx,dy,dz,heading, E extra columns such as velocity, class 1..num_classes LAST; zero rows =
    padding); feature_map_size (H, W). Returns heatmaps [B,nc,H,W], target_boxes [B,K,8 + E] (the extras behind the 8 regression
    targets, center_head.py:115,154-155), inds [B,K] i64, masks [B,K] i64 (K = num_max_objs) -- assign_target_of_single_head for
    every sample."""
    B, M, C = gt_boxes.shape
    assert C >= 8, "gt_boxes rows are (x, y, z, dx, dy, dz, heading, [extras ...], class)"
    H, W = int(feature_map_size[0]), int(feature_map_size[1])
    dev = gt_boxes.device
    K = num_max_objs
    # the reference first keeps this head's boxes (class >= 1: padding rows are 'bg', center_head.py:180-196), THEN walks the
    # first NUM_MAX_OBJS of that filtered list (l.113): compact the kept rows to the front, stably, before truncating
    if M > 0:
        drop = (gt_boxes[..., -1] < 1).to(torch.int8)
        order = torch.sort(drop, dim=1, stable=True)[1]
        gt_boxes = torch.gather(gt_boxes, 1, order.unsqueeze(-1).expand(-1, -1, gt_boxes.shape[2]))
    g = gt_boxes[:, :K].float()
    Mk = g.shape[1]
    x, y, z = g[..., 0], g[..., 1], g[..., 2]
    cx = ((x - point_cloud_range[0]) / voxel_size[0] / feature_map_stride).clamp(min=0, max=W - 0.5)
    cy = ((y - point_cloud_range[1]) / voxel_size[1] / feature_map_stride).clamp(min=0, max=H - 0.5)
    cxi, cyi = cx.int(), cy.int()
    dx = g[..., 3] / voxel_size[0] / feature_map_stride
    dy = g[..., 4] / voxel_size[1] / feature_map_stride
    valid = (dx > 0) & (dy > 0) & (g[..., -1] >= 1)
    radius = torch.clamp_min(gaussian_radius(dx.clamp_min(1e-6), dy.clamp_min(1e-6), min_overlap=gaussian_overlap).int(), min_radius)
    radius = torch.where(valid, radius, torch.zeros_like(radius))

    heat = torch.zeros((B, num_classes, H, W), dtype=torch.float32, device=dev)
    if Mk > 0 and bool(valid.any()):
        R = int(radius.max().item())
        off = torch.arange(-R, R + 1, device=dev)
        oy, ox = torch.meshgrid(off, off, indexing="ij")                       # [D, D]
        r = radius[..., None, None]                                             # [B, Mk, 1, 1]
        sigma = (2 * r.double() + 1) / 6
        gauss = torch.exp(-(ox.double() ** 2 + oy.double() ** 2)[None, None] / (2 * sigma * sigma)).float()
        px = cxi[..., None, None] + ox[None, None]
        py = cyi[..., None, None] + oy[None, None]
        inside = (ox.abs()[None, None] <= r) & (oy.abs()[None, None] <= r) & (px >= 0) & (px < W) & (py >= 0) & (py < H) \
            & valid[..., None, None]
        cls = (g[..., -1].long() - 1).clamp(0, num_classes - 1)[..., None, None]
        b_idx = torch.arange(B, device=dev)[:, None, None, None]
        lin = ((b_idx * num_classes + cls) * H + py.clamp(0, H - 1).long()) * W + px.clamp(0, W - 1).long()
        vals = torch.where(inside, gauss, torch.zeros_like(gauss))
        heat.view(-1).scatter_reduce_(0, lin.reshape(-1), vals.reshape(-1), reduce="amax", include_self=True)

    target = torch.zeros((B, K, C), dtype=torch.float32, device=dev)
    inds = torch.zeros((B, K), dtype=torch.int64, device=dev)
    masks = torch.zeros((B, K), dtype=torch.int64, device=dev)
    if Mk > 0:
        t = torch.stack([cx - cxi.float(), cy - cyi.float(), z, g[..., 3].clamp_min(1e-12).log(), g[..., 4].clamp_min(1e-12).log(),
                         g[..., 5].clamp_min(1e-12).log(), torch.cos(g[..., 6]), torch.sin(g[..., 6])], dim=-1)
        if C > 8:
            t = torch.cat([t, g[..., 7:-1]], dim=-1)
        v = valid[..., None].float()
        target[:, :Mk] = t * v
        inds[:, :Mk] = (cyi.long() * W + cxi.long()) * valid.long()
        masks[:, :Mk] = valid.long()
    return heat, target, inds, masks


def neg_loss_cornernet(pred, gt):
    """loss_utils.py:265-300 (mask=None)."""
    pos_inds = gt.eq(1).float()
    neg_inds = gt.lt(1).float()
    neg_weights = torch.pow(1 - gt, 4)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos_inds).sum()
    neg_loss = (torch.log(1 - pred) * torch.pow(pred, 2) * neg_weights * neg_inds).sum()
    num_pos = pos_inds.sum()
    if num_pos == 0:
        return -neg_loss
    return -(pos_loss + neg_loss) / num_pos


def reg_loss(pred, target, mask):
    """loss_utils.py:315-350: pred/target [B,K,D], mask [B,K] -> per-dimension L1 [D]."""
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float() * (~torch.isnan(target)).float()
    loss = torch.abs(pred * m - target * m).sum(dim=(0, 1))
    return loss / torch.clamp_min(num, min=1.0)


def center_head_loss(head_rows, batch, h, w, heatmaps, target_boxes, inds, masks, num_classes=3, hm_col=8,
                     code_weights=None, loc_weight=2.0, cls_weight=1.0):
    """CenterHead.get_loss (center_head.py:225-250) on channels-last head rows [B*H*W, ld]:
    columns 0..7 = center(2), center_z, dim(3), rot(2) in HEAD_ORDER, columns hm_col.. = hm logits."""
    hm = head_rows[:, hm_col:hm_col + num_classes]
    pred_hm = torch.clamp(hm.sigmoid(), min=1e-4, max=1 - 1e-4)                          # center_head.py:221-223
    gt_rows = heatmaps.permute(0, 2, 3, 1).reshape(batch * h * w, num_classes)
    hm_loss = neg_loss_cornernet(pred_hm, gt_rows) * cls_weight
    base = (torch.arange(batch, device=head_rows.device) * (h * w))[:, None]
    pred = head_rows[(base + inds).reshape(-1), 0:8].view(batch, -1, 8)                   # _transpose_and_gather_feat
    rl = reg_loss(pred, target_boxes, masks)
    cw = rl.new_tensor(code_weights if code_weights is not None else [1.0] * 8)
    loc_loss = (rl * cw).sum() * loc_weight
    return hm_loss + loc_loss, {"hm_loss": hm_loss.detach(), "loc_loss": loc_loss.detach()}
