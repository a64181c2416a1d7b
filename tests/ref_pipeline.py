"""The reference's module graph executed on the CPU ORACLE (test infrastructure only).

Walks the same state_dict (reference names / layouts) through oracle primitives the way the
reference modules do -- separate conv, bias, eval BatchNorm, ReLU, residual; NCHW dense tensors --
so that the HIP engine's fusions, weight repacking and channels-last layout are checked against an
unfused restatement:
  MeanVFE (mean_vfe.py:41-43) -> VoxelResBackBone8x (spconv_backbone.py:502-558)
  -> HeightCompression (height_compression.py:136-138) -> BaseBEVBackbone (base_bev_backbone.py:85-122)
  -> CenterHead (center_head.py:323-338, 252-303) -> class_agnostic_nms (model_nms_utils.py:115-134)
"""
import numpy as np

DOWN = {
    "conv2": ([3, 3, 3], [2, 2, 2], [1, 1, 1]),
    "conv3": ([3, 3, 3], [2, 2, 2], [1, 1, 1]),
    "conv4": ([3, 3, 3], [2, 2, 2], [0, 1, 1]),
    "conv_out": ([3, 1, 1], [2, 1, 1], [0, 0, 0]),
}


def _np(sd, k):
    return sd[k].detach().cpu().numpy().astype(np.float32)


def bn_rows(x, sd, name, eps):
    g, b, m, v = (_np(sd, name + s) for s in (".weight", ".bias", ".running_mean", ".running_var"))
    inv = (1.0 / np.sqrt(v + np.float32(eps))).astype(np.float32)
    return ((x - m) * inv * g + b).astype(np.float32)


def relu(x):
    return np.maximum(x, np.float32(0))


def voxelize_batch(o, cfg, points_list):
    feats, coords = [], []
    for b, pts in enumerate(points_list):
        v, c, n = o.voxelize(pts, cfg.voxel_size, cfg.point_cloud_range, cfg.max_points_per_voxel, cfg.max_voxels)
        feats.append(o.mean_vfe(v, n))
        coords.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    return np.concatenate(feats), np.concatenate(coords)


def backbone3d(o, cfg, sd, feats, coords, batch):
    p = "backbone_3d."
    eps = 1e-3
    shape = cfg.sparse_shape
    K3 = [3, 3, 3]

    def conv(name, x, nbr):
        w = _np(sd, name + ".weight")
        b = _np(sd, name + ".bias") if (name + ".bias") in sd else None
        return o.sparse_conv(x, w, b, nbr)

    def block(name, x, nbr):
        y = relu(bn_rows(conv(name + ".conv1", x, nbr), sd, name + ".bn1", eps))
        y = bn_rows(conv(name + ".conv2", y, nbr), sd, name + ".bn2", eps)
        return relu(y + x)

    nbr = o.subm_rulebook(coords, batch, shape, K3)
    x = relu(bn_rows(conv(p + "conv_input.0", feats, nbr), sd, p + "conv_input.1", eps))
    x = block(p + "conv1.0", x, nbr)
    x = block(p + "conv1.1", x, nbr)
    levels = {"x_conv1": (x, coords, shape)}
    for i, stage in enumerate(["conv2", "conv3", "conv4"], start=2):
        k, s, pd = DOWN[stage]
        out_idx = o.conv_outset(coords, batch, shape, k, s, pd)
        nbr_dn = o.conv_rulebook(coords, out_idx, batch, shape, k, s, pd)
        x = relu(bn_rows(conv(p + stage + ".0.0", x, nbr_dn), sd, p + stage + ".0.1", eps))
        shape = o.conv_out_shape(shape, k, s, pd)
        coords = out_idx
        nbr = o.subm_rulebook(coords, batch, shape, K3)
        x = block(p + stage + ".1", x, nbr)
        x = block(p + stage + ".2", x, nbr)
        levels["x_conv%d" % i] = (x, coords, shape)
    k, s, pd = DOWN["conv_out"]
    out_idx = o.conv_outset(coords, batch, shape, k, s, pd)
    nbr_dn = o.conv_rulebook(coords, out_idx, batch, shape, k, s, pd)
    x = relu(bn_rows(conv(p + "conv_out.0", x, nbr_dn), sd, p + "conv_out.1", eps))
    return levels, (x, out_idx, o.conv_out_shape(shape, k, s, pd))


def bn2d(o, x, sd, name, eps, relu_=True):
    return o.bn_relu(x, _np(sd, name + ".weight"), _np(sd, name + ".bias"), _np(sd, name + ".running_mean"),
                     _np(sd, name + ".running_var"), eps, relu_)


def bev_backbone(o, cfg, sd, x):
    p = "backbone_2d."
    ups = []
    for lvl in range(len(cfg.bev_layer_nums)):
        x = o.conv2d(x, _np(sd, p + "blocks.%d.1.weight" % lvl), None, cfg.bev_layer_strides[lvl], 1)
        x = bn2d(o, x, sd, p + "blocks.%d.2" % lvl, 1e-3)
        for k in range(cfg.bev_layer_nums[lvl]):
            x = o.conv2d(x, _np(sd, p + "blocks.%d.%d.weight" % (lvl, 4 + 3 * k)), None, 1, 1)
            x = bn2d(o, x, sd, p + "blocks.%d.%d" % (lvl, 5 + 3 * k), 1e-3)
        u = o.deconv2d(x, _np(sd, p + "deblocks.%d.0.weight" % lvl), cfg.bev_upsample_strides[lvl])
        ups.append(bn2d(o, u, sd, p + "deblocks.%d.1" % lvl, 1e-3))
    return np.concatenate(ups, 1)


def center_head(o, cfg, sd, x):
    p = "dense_head."
    x = o.conv2d(x, _np(sd, p + "shared_conv.0.weight"), _np(sd, p + "shared_conv.0.bias"), 1, 1)
    x = bn2d(o, x, sd, p + "shared_conv.1", 1e-5)
    out = {}
    for name in cfg.head_names():
        q = p + "heads_list.0.%s." % name
        h = o.conv2d(x, _np(sd, q + "0.0.weight"), _np(sd, q + "0.0.bias"), 1, 1)
        h = bn2d(o, h, sd, q + "0.1", 1e-5)
        out[name] = o.conv2d(h, _np(sd, q + "1.weight"), _np(sd, q + "1.bias"), 1, 1)
    return out


def decode_nms(o, cfg, heads, b):
    boxes, scores, labels = o.center_decode(
        heads["hm"][b], heads["center"][b], heads["center_z"][b], heads["dim"][b], heads["rot"][b],
        cfg.max_obj_per_sample, float(cfg.feature_map_stride), cfg.voxel_size[:2], cfg.point_cloud_range[:2],
        cfg.post_center_limit_range, cfg.score_thresh)
    order = np.argsort(-scores, kind="stable")[:cfg.nms_pre_maxsize]
    keep = o.nms(boxes[order], cfg.nms_thresh)
    sel = order[keep][:cfg.nms_post_maxsize]
    return dict(pred_boxes=boxes[sel], pred_scores=scores[sel], pred_labels=labels[sel].astype(np.int64) + 1,
                decoded=(boxes, scores, labels))


def forward(o, cfg, sd, points_list):
    batch = len(points_list)
    feats, coords = voxelize_batch(o, cfg, points_list)
    levels, (x, idx, shape) = backbone3d(o, cfg, sd, feats, coords, batch)
    spatial = o.densify(x, idx, batch, shape)                       # (B, C*D, H, W)
    bev = bev_backbone(o, cfg, sd, spatial)
    heads = center_head(o, cfg, sd, bev)
    results = [decode_nms(o, cfg, heads, b) for b in range(batch)]
    inter = dict(voxel_features=feats, voxel_coords=coords, levels=levels, encoded=(x, idx, shape),
                 spatial_features=spatial, bev=bev, heads=heads)
    return results, inter
