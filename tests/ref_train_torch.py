"""Training-mode reference of the hot path as a torch-CPU float64 AUTOGRAD graph (test infrastructure).

The reference's modules executed the way the reference runs them in `model.train()`:
  VoxelResBackBone8x (spconv_backbone.py:502-558; SparseBasicBlock l.120-136) with the sparse convs in
  their defining gather-matmul form over the ORACLE's rulebooks, nn.BatchNorm1d on batch statistics;
  HeightCompression (height_compression.py:136-138: dense NCHW, channel = c*D + z);
  BaseBEVBackbone (base_bev_backbone.py:85-122) and CenterHead / SeparateHead (center_head.py:11-45,
  73-94, 323-330) through F.conv2d / F.conv_transpose2d / F.batch_norm(training=True) on the
  reference's own weight layouts; loss = cpd_amd.center_loss (itself pinned on reference goldens).
`loss.backward()` then gives d(loss)/d(parameter) under the reference's state_dict names, which
tests/test_gpu_train.py compares with the HIP trainer's hand-written backward.

ReLU kinks: with ~2e7 activations a handful sit within fp32 rounding of zero, and an fp32 run and
this float64 run take different branches there -- a legitimate O(upstream gradient) difference in
d(loss)/d(weight) that says nothing about the kernels. `masks` (layer name -> bool tensor in the
reference's layout) pins every ReLU to the branch the run under test took, so the comparison is of
the same piecewise-linear function and tolerances can stay at rounding level.
"""
import numpy as np
import torch
import torch.nn.functional as F

from cpd_amd import center_loss
from ref_pipeline import DOWN, voxelize_batch

F64 = torch.float64


def _gconv(x, w_ref, nbr, bias=None):
    """x [n_in, ci]; w_ref (co, kd, kh, kw, ci) spconv layout; nbr [kv, n_out] (-1 = no neighbour)."""
    co, ci = w_ref.shape[0], w_ref.shape[-1]
    w = w_ref.reshape(co, -1, ci).permute(1, 2, 0)              # [kv, ci, co]
    xp = torch.cat([x, x.new_zeros(1, ci)])
    idx = torch.as_tensor(np.where(nbr < 0, x.shape[0], nbr), dtype=torch.long)
    out = x.new_zeros(nbr.shape[1], co)
    for t in range(w.shape[0]):
        out = out + xp[idx[t]] @ w[t]
    if bias is not None:
        out = out + bias
    return out


class _Relu:
    def __init__(self, masks):
        self.masks = masks

    def __call__(self, v, key):
        if self.masks is None or key not in self.masks:
            return F.relu(v)
        m = self.masks[key]
        assert m.shape == v.shape, (key, m.shape, v.shape)
        return v * m.to(v.dtype)


def _bn_rows(z, P, name, eps):
    return F.batch_norm(z, None, None, P[name + ".weight"], P[name + ".bias"], training=True, eps=eps)


def make_leaves(sd, dtype=F64):
    """Leaf copies (float64 by default) of every trainable tensor of a reference-layout state_dict."""
    P = {}
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
            continue
        P[k] = v.detach().cpu().to(dtype).clone().requires_grad_(True)
    return P


def forward_loss(o, cfg, P, points_list, gt_boxes, num_max_objs=500, taps=None, masks=None):
    """Returns (loss, parts, head maps dict NCHW) with autograd history back to the leaves `P`
    (computed in the leaves' dtype: float64 = the reference, float32 = what torch itself loses)."""
    F64 = next(iter(P.values())).dtype
    relu = _Relu(masks)
    batch = len(points_list)
    feats, coords = voxelize_batch(o, cfg, [np.asarray(p, dtype=np.float32) for p in points_list])
    x = torch.as_tensor(feats, dtype=F64)
    p = "backbone_3d."
    eps = 1e-3
    shape = cfg.sparse_shape
    K3 = [3, 3, 3]

    def block(name, x, nbr):
        z1 = _gconv(x, P[name + ".conv1.weight"], nbr, P.get(name + ".conv1.bias"))
        y = relu(_bn_rows(z1, P, name + ".bn1", eps), name + ".conv1")
        z2 = _gconv(y, P[name + ".conv2.weight"], nbr, P.get(name + ".conv2.bias"))
        out = relu(_bn_rows(z2, P, name + ".bn2", eps) + x, name + ".conv2")
        if taps is not None:
            taps[name + ".conv1"] = (z1, y)
            taps[name + ".conv2"] = (z2, out)
        return out

    nbr = o.subm_rulebook(coords, batch, shape, K3)
    x = relu(_bn_rows(_gconv(x, P[p + "conv_input.0.weight"], nbr), P, p + "conv_input.1", eps), p + "conv_input.0")
    x = block(p + "conv1.0", x, nbr)
    x = block(p + "conv1.1", x, nbr)
    for stage in ["conv2", "conv3", "conv4", "conv_out"]:
        k, s, pd = DOWN[stage]
        out_idx = o.conv_outset(coords, batch, shape, k, s, pd)
        nbr_dn = o.conv_rulebook(coords, out_idx, batch, shape, k, s, pd)
        shape = o.conv_out_shape(shape, k, s, pd)
        coords = out_idx
        if stage == "conv_out":
            x = relu(_bn_rows(_gconv(x, P[p + "conv_out.0.weight"], nbr_dn), P, p + "conv_out.1", eps), p + "conv_out.0")
            break
        x = relu(_bn_rows(_gconv(x, P[p + stage + ".0.0.weight"], nbr_dn), P, p + stage + ".0.1", eps), p + stage + ".0.0")
        nbr = o.subm_rulebook(coords, batch, shape, K3)
        x = block(p + stage + ".1", x, nbr)
        x = block(p + stage + ".2", x, nbr)

    # HeightCompression: dense (B, C, D, H, W) -> view (B, C*D, H, W)
    d, h, w = shape
    C = x.shape[1]
    ci = torch.as_tensor(coords, dtype=torch.long)
    dense = x.new_zeros(batch, C, d, h, w)
    dense = _scatter_dense(dense, ci, x)
    sp = dense.view(batch, C * d, h, w)

    # BaseBEVBackbone
    p = "backbone_2d."
    ups = []
    xx = sp
    for lvl in range(len(cfg.bev_layer_nums)):
        stride = cfg.bev_layer_strides[lvl]
        xx = F.pad(xx, (1, 1, 1, 1))                                            # nn.ZeroPad2d(1), l.33
        xx = F.conv2d(xx, P[p + "blocks.%d.1.weight" % lvl], None, stride=stride, padding=0)
        xx = relu(F.batch_norm(xx, None, None, P[p + "blocks.%d.2.weight" % lvl], P[p + "blocks.%d.2.bias" % lvl],
                               training=True, eps=1e-3), p + "blocks.%d.1" % lvl)
        for k in range(cfg.bev_layer_nums[lvl]):
            xx = F.conv2d(xx, P[p + "blocks.%d.%d.weight" % (lvl, 4 + 3 * k)], None, padding=1)
            xx = relu(F.batch_norm(xx, None, None, P[p + "blocks.%d.%d.weight" % (lvl, 5 + 3 * k)],
                                   P[p + "blocks.%d.%d.bias" % (lvl, 5 + 3 * k)], training=True, eps=1e-3),
                      p + "blocks.%d.%d" % (lvl, 4 + 3 * k))
        u = cfg.bev_upsample_strides[lvl]
        up = F.conv_transpose2d(xx, P[p + "deblocks.%d.0.weight" % lvl], None, stride=u)
        up = relu(F.batch_norm(up, None, None, P[p + "deblocks.%d.1.weight" % lvl], P[p + "deblocks.%d.1.bias" % lvl],
                               training=True, eps=1e-3), p + "deblocks.%d.0" % lvl)
        ups.append(up)
    cat = torch.cat(ups, dim=1)

    # CenterHead
    p = "dense_head."
    s = F.conv2d(cat, P[p + "shared_conv.0.weight"], P[p + "shared_conv.0.bias"], padding=1)
    s = relu(F.batch_norm(s, None, None, P[p + "shared_conv.1.weight"], P[p + "shared_conv.1.bias"], training=True,
                          eps=1e-5), p + "shared_conv.0")
    maps = {}
    for name in cfg.head_names():
        q = p + "heads_list.0.%s." % name
        y = F.conv2d(s, P[q + "0.0.weight"], P[q + "0.0.bias"], padding=1)
        y = relu(F.batch_norm(y, None, None, P[q + "0.1.weight"], P[q + "0.1.bias"], training=True, eps=1e-5), q + "0.0")
        maps[name] = F.conv2d(y, P[q + "1.weight"], P[q + "1.bias"], padding=1)

    # loss on channels-last rows in HEAD_ORDER + hm (the layout center_loss is pinned on)
    order = list(cfg.head_order) + ["hm"]
    rows = torch.cat([maps[n] for n in order], dim=1).permute(0, 2, 3, 1).reshape(batch * h * w, -1)
    hm_col = sum(cfg.head_channels[n] for n in cfg.head_order)
    heat, tgt, inds, masks = center_loss.assign_targets(
        torch.as_tensor(gt_boxes, dtype=torch.float32), (h, w), cfg.point_cloud_range, cfg.voxel_size, cfg.num_class,
        cfg.feature_map_stride, num_max_objs=num_max_objs)
    loss, parts = center_loss.center_head_loss(rows, batch, h, w, heat.to(F64), tgt.to(F64), inds, masks, cfg.num_class,
                                               hm_col=hm_col)
    return loss, parts, maps


def _scatter_dense(dense, ci, x):
    b, c, d, h, w = dense.shape
    flat = dense.permute(0, 2, 3, 4, 1).reshape(b * d * h * w, c)
    lin = ((ci[:, 0] * d + ci[:, 1]) * h + ci[:, 2]) * w + ci[:, 3]
    flat = flat.index_copy(0, lin, x)
    return flat.view(b, d, h, w, c).permute(0, 4, 1, 2, 3).contiguous()
