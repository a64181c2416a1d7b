"""The reference's SECOND stage (VoxelRCNNHead eval branch + post_processing) executed on the CPU ORACLE and numpy
(test infrastructure only). Un-fused, module by module, the way the reference does it:

  roi_grid_pool                          cpd/models/roi_heads/voxel_rcnn_head.py:186-273
    get_global_grid_points_of_roi        voxel_rcnn_head.py:365-386 (+ common_utils.rotate_points_along_z, common_utils.py:35-57)
    get_voxel_centers                    cpd/utils/common_utils.py:66-82
    generate_voxel2pinds                 cpd/utils/spconv_utils.py:14-21                      -> oracle.voxel2pinds
    NeighborVoxelSAModuleMSG.forward     cpd/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:70-128
      voxel_query                        pointnet2_stack/src/voxel_query_gpu.cu:10-87          -> oracle.voxel_query
      grouping                           pointnet2_stack/src/group_points_gpu.cu:71            (numpy fancy indexing == oracle.group_points)
  shared_fc / cls / reg stacks           voxel_rcnn_head.py:67-93, 694-705 (Linear, eval BatchNorm1d, ReLU; Dropout is identity in eval)
  generate_predicted_boxes               cpd/models/roi_heads/roi_head_template.py:269-299     -> oracle.anchor_decode + rotation
  post_processing                        cpd/models/detectors/detector3d_template.py:222-343   -> oracle.nms

fp32 where the reference is fp32 and a DECISION hangs on the value (grid points, cell coordinates); the matrix products run in
float64 (the checker must be at least as accurate as the thing it checks)."""
import numpy as np

F = np.float32


def _sd(sd, k):
    v = sd[k]
    return (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v))


def fold_bn(sd, name, eps=1e-5):
    g, b, m, v = (_sd(sd, name + s).astype(np.float64) for s in (".weight", ".bias", ".running_mean", ".running_var"))
    s = g / np.sqrt(v + eps)
    return s, b - m * s


def grid_points(rois, grid_size):
    """(B*N, G^3, 3) fp32 global grid points and the local ones; the reference's operation order in fp32."""
    r = rois.reshape(-1, rois.shape[-1]).astype(F)
    g = grid_size
    idx = np.stack(np.meshgrid(np.arange(g), np.arange(g), np.arange(g), indexing="ij"), -1).reshape(-1, 3).astype(F)   # ones(G,G,G).nonzero()
    size = r[:, None, 3:6]
    local = ((idx[None] + F(0.5)) / F(g) * size - size / F(2)).astype(F)
    ca, sa = np.cos(r[:, 6]).astype(F), np.sin(r[:, 6]).astype(F)
    z, o = np.zeros_like(ca), np.ones_like(ca)
    rot = np.stack([ca, sa, z, -sa, ca, z, z, z, o], 1).reshape(-1, 3, 3).astype(F)
    out = np.einsum("nki,nij->nkj", local, rot).astype(F) + r[:, None, 0:3]
    return out.astype(F), local


def voxel_centers(coords_zyx, stride, voxel_size, pc_range):
    c = coords_zyx[:, [2, 1, 0]].astype(F)
    vs = (np.asarray(voxel_size, F) * F(stride)).astype(F)
    return ((c + F(0.5)) * vs + np.asarray(pc_range[:3], F)).astype(F)


def pool_scale(o, sd, prefix, k, feats, xyz, v2p, new_xyz, new_coords_bzyx, query_range, radius, nsample, idx=None):
    """one scale of NeighborVoxelSAModuleMSG (voxel_pool_modules.py:86-128), eval BatchNorm; -> (M, C2) float64, idx"""
    w_in = _sd(sd, prefix + "mlps_in.%d.0.weight" % k)[:, :, 0].astype(np.float64)               # (C1, C0)
    s_in, t_in = fold_bn(sd, prefix + "mlps_in.%d.1" % k)
    fin = feats.astype(np.float64) @ w_in.T * s_in + t_in                                         # mlps_in: conv + BN (no ReLU)
    if idx is None:
        idx = o.voxel_query(query_range, radius, nsample, xyz, new_xyz, new_coords_bzyx, v2p)
    empty = idx[:, 0] == -1
    idx = idx.copy()
    idx[empty] = 0
    gf = fin[idx]                                                                                 # (M, ns, C1)
    gx = (xyz[idx] - new_xyz[:, None, :]).astype(np.float64)                                      # fp32 difference, as the reference forms it
    gf[empty] = 0
    gx[empty] = 0
    w_pos = _sd(sd, prefix + "mlps_pos.%d.0.weight" % k)[:, :, 0, 0].astype(np.float64)          # (C1, 3)
    s_p, t_p = fold_bn(sd, prefix + "mlps_pos.%d.1" % k)
    pos = gx @ w_pos.T * s_p + t_p
    x = np.maximum(gf + pos, 0).max(axis=1)                                                       # ReLU, max over the samples
    w_out = _sd(sd, prefix + "mlps_out.%d.0.weight" % k)[:, :, 0].astype(np.float64)
    s_o, t_o = fold_bn(sd, prefix + "mlps_out.%d.1" % k)
    return np.maximum(x @ w_out.T * s_o + t_o, 0), idx, empty


def roi_grid_pool(o, sd, roi_cfg, rois, levels, strides, voxel_size, pc_range, batch, grid_xyz=None, prefix="roi_head."):
    pool = roi_cfg["ROI_GRID_POOL"]
    g = pool["GRID_SIZE"]
    if grid_xyz is None:
        grid_xyz, _ = grid_points(rois, g)
    gxyz = grid_xyz.reshape(batch, -1, 3).astype(F)
    lo, vs = np.asarray(pc_range[:3], F), np.asarray(voxel_size, F)
    gc = np.floor((gxyz - lo) / vs).astype(F)                                                     # torch `//` on floats = floor of the fp32 quotient
    m_per = gxyz.shape[1]
    bidx = np.repeat(np.arange(batch, dtype=np.int32), m_per)
    new_xyz = np.ascontiguousarray(gxyz.reshape(-1, 3))
    outs, queries = [], {}
    for li, name in enumerate(pool["FEATURES_SOURCE"]):
        feats, coords, shape = levels[name]
        stride = strides[name]
        xyz = np.ascontiguousarray(voxel_centers(coords[:, 1:4], stride, voxel_size, pc_range))
        cur = np.floor(gc / F(stride)).astype(np.int32).reshape(-1, 3)                            # (x, y, z) cells at this level
        nc = np.ascontiguousarray(np.concatenate([bidx[:, None], cur[:, [2, 1, 0]]], 1).astype(np.int32))   # (b, z, y, x)
        v2p = o.voxel2pinds(np.ascontiguousarray(coords.astype(np.int32)), batch, [int(s) for s in shape])
        lc = pool["POOL_LAYERS"][name]
        for k in range(len(lc["NSAMPLE"])):
            y, idx, empty = pool_scale(o, sd, prefix + "roi_grid_pool_layers.%d." % li, k, feats, xyz, v2p, new_xyz, nc,
                                       lc["QUERY_RANGES"][k], float(lc["POOL_RADIUS"][k]), int(lc["NSAMPLE"][k]))
            outs.append(y)
            queries[(name, k)] = (idx, empty)
    pooled = np.concatenate(outs, 1)                                                              # (B*N*G^3, sum C2)
    return pooled.reshape(-1, g ** 3, pooled.shape[-1]), queries


def fc_stack(sd, prefix, x):
    """the nn.Sequential of voxel_rcnn_head.py:67-93 by its state-dict indices: Linear [+ BatchNorm1d + ReLU] (Dropout = identity in eval); float64"""
    x = x.astype(np.float64)
    for i in range(32):
        kw = prefix + "%d.weight" % i
        if kw not in sd or _sd(sd, kw).ndim != 2:
            continue
        x = x @ _sd(sd, kw).astype(np.float64).T
        if prefix + "%d.bias" % i in sd:
            x = x + _sd(sd, prefix + "%d.bias" % i).astype(np.float64)
        if prefix + "%d.running_mean" % (i + 1) in sd:
            s, t = fold_bn(sd, prefix + "%d" % (i + 1))
            x = np.maximum(x * s + t, 0)
    return x


def predicted_boxes(o, rois, cls, reg):
    """roi_head_template.py:269-299 (ResidualCoder.decode_torch on the RoI moved to the origin, rotate by its heading, translate)"""
    flat = rois.reshape(-1, rois.shape[-1])[:, :7].astype(F)
    local = flat.copy()
    local[:, 0:3] = 0
    dec = o.anchor_decode(np.ascontiguousarray(reg.astype(F)).reshape(1, -1, 7), np.ascontiguousarray(local))[0]
    ca, sa = np.cos(flat[:, 6]).astype(F), np.sin(flat[:, 6]).astype(F)
    x = dec[:, 0] * ca - dec[:, 1] * sa
    y = dec[:, 0] * sa + dec[:, 1] * ca
    out = np.concatenate([(x + flat[:, 0])[:, None], (y + flat[:, 1])[:, None], (dec[:, 2] + flat[:, 2])[:, None], dec[:, 3:]], 1).astype(F)
    return cls.reshape(rois.shape[0], rois.shape[1], -1), out.reshape(rois.shape[0], rois.shape[1], 7)


def post_processing(o, post_cfg, boxes, cls, roi_labels, sigmoid_dtype=np.float64):
    """detector3d_template.py:222-343 with MULTI_CLASSES_NMS False and has_class_labels: per frame sigmoid, threshold, descending
    score order (ties -> lower index), rotated NMS, labels from the RoIs."""
    nms = post_cfg["NMS_CONFIG"]
    out = []
    for b in range(boxes.shape[0]):
        x = cls[b].astype(sigmoid_dtype)
        s = (sigmoid_dtype(1.0) / (sigmoid_dtype(1.0) + np.exp(-x))).max(-1)
        ok = np.nonzero(s >= post_cfg["SCORE_THRESH"])[0]
        order = ok[np.argsort(-s[ok], kind="stable")][:int(nms["NMS_PRE_MAXSIZE"])]
        keep = o.nms(np.ascontiguousarray(boxes[b][order].astype(F)), float(nms["NMS_THRESH"]))
        sel = order[keep][:int(nms["NMS_POST_MAXSIZE"])]
        out.append(dict(pred_boxes=boxes[b][sel], pred_scores=s[sel].astype(F), pred_labels=roi_labels[b][sel], selected=sel))
    return out


def second_stage(o, cfg, roi_cfg, post_cfg, sd, rois, roi_labels, levels, batch, grid_xyz=None):
    strides = {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}
    pooled, queries = roi_grid_pool(o, sd, roi_cfg, rois, levels, strides, cfg.voxel_size, cfg.point_cloud_range, batch, grid_xyz)
    x = pooled.reshape(pooled.shape[0], -1)
    shared = fc_stack(sd, "roi_head.shared_fc_layers.", x)
    cls = fc_stack(sd, "roi_head.cls_layers.", shared)
    reg = fc_stack(sd, "roi_head.reg_layers.", shared)
    cls_b, boxes = predicted_boxes(o, rois, cls, reg)
    final = post_processing(o, post_cfg, boxes, cls_b, roi_labels)
    return final, dict(pooled=pooled, shared=shared, rcnn_cls=cls, rcnn_reg=reg, batch_box_preds=boxes, batch_cls_preds=cls_b, queries=queries)
