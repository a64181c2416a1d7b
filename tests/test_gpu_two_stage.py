"""The two-stage detector of the shipped config (voxel_rcnn_cproto_center.yaml:13 `NAME: VoxelRCNN`): the fused engine
(cpd_amd/two_stage.py) against the drop-in module composition (cpd_amd.models.VoxelRCNN: CenterPoint modules -> CenterHead rois ->
VoxelRCNNProtoHead eval branch -> post_processing), whose pieces are pinned on the reference's own goldens elsewhere
(test_gpu_roi_pool.py: voxel_rcnn_head.npz / roi_pool.npz; test_gpu_decode_nms.py: nms goldens; test_gpu_pipeline.py: stage one vs
the oracle), and post_processing against a plain restatement of detector3d_template.py:222-343."""
import numpy as np
import pytest
import torch

from cpd_amd import models, ops
from cpd_amd.synthetic import waymo_cloud

pytestmark = pytest.mark.gpu


def _model(seed=3):
    cfg = models.waymo_voxel_rcnn_cfg()
    cfg.BACKBONE_2D.NUM_FILTERS, cfg.BACKBONE_2D.NUM_UPSAMPLE_FILTERS, cfg.BACKBONE_2D.LAYER_NUMS = [64, 128], [128, 128], [2, 2]
    cfg.DENSE_HEAD.POST_PROCESSING.POST_CENTER_LIMIT_RANGE = [-20, -20, -2, 20, 20, 4]
    cfg.DENSE_HEAD.POST_PROCESSING.MAX_OBJ_PER_SAMPLE = 100
    torch.manual_seed(seed)
    net = models.VoxelRCNN(cfg, point_cloud_range=[-20.0, -20.0, -2.0, 20.0, 20.0, 4.0]).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.75, 1.25)
                m.weight.uniform_(0.75, 1.25); m.bias.normal_(0, 0.1)
        for stack in (net.roi_head.cls_layers, net.roi_head.reg_layers):       # the reference's N(0, 0.01) output layers give constant
            stack[-1].weight.normal_(0, 0.3); stack[-1].bias.normal_(0, 0.3)   # scores: spread them, so that the final NMS has work
    return net


def _clouds():
    out = []
    for s_ in (0, 1, 2):
        p = waymo_cloud(s_, n_points=40000 + 5000 * s_)
        p[:, :2] *= 0.3
        out.append(torch.from_numpy(p).cuda())
    return out


def _batch_dict(net, clouds):
    vox = ops.Voxelizer(net.voxel_size, net.point_cloud_range, 5, 5, 1000000)
    _, coords, _, feats, nvox = vox.batch(clouds)
    n = int(nvox[len(clouds)])
    return {"voxel_features": feats[:n].clone(), "voxel_coords": coords[:n].float(), "batch_size": len(clouds)}


def _match(a, b, atol):
    """every box of a has a partner in b (same geometry to atol), one to one"""
    assert a["pred_boxes"].shape == b["pred_boxes"].shape, (a["pred_boxes"].shape, b["pred_boxes"].shape)
    x, y = a["pred_boxes"].cpu().numpy(), b["pred_boxes"].cpu().numpy()
    if len(x) == 0:
        return
    j = np.abs(x[:, None, :] - y[None, :, :]).max(-1).argmin(1)
    assert sorted(j.tolist()) == list(range(len(y)))
    np.testing.assert_allclose(x, y[j], atol=atol, rtol=1e-4)
    np.testing.assert_allclose(a["pred_scores"].cpu().numpy(), b["pred_scores"].cpu().numpy()[j], atol=atol)
    np.testing.assert_array_equal(a["pred_labels"].cpu().numpy(), b["pred_labels"].cpu().numpy()[j])


def test_two_stage_engine_matches_the_module_composition(hip):
    from cpd_amd import spconv as sp
    sp.install(conv_math="f32")
    net = _model()
    clouds = _clouds()
    with torch.no_grad():
        want, _, bd = net(_batch_dict(net, clouds))
    assert bd["rois"].shape[1] == max(1, max(int((bd["roi_scores"][b] > 0).sum()) for b in range(3)))
    ecfg = net.to_engine_config()
    ecfg.conv_math = "f32"
    eng = net.to_engine()
    eng_f32 = type(eng)(ecfg, net.model_cfg.ROI_HEAD, net.model_cfg.POST_PROCESSING, {k: v.detach().cpu() for k, v in net.state_dict().items()})
    got, it = eng_f32.forward(clouds, return_intermediates=True)
    assert sum(len(g["pred_boxes"]) for g in got) > 10, "nothing survived: the comparison would be vacuous"
    assert it["rois"].shape == bd["rois"].shape
    np.testing.assert_allclose(it["rois"].cpu().numpy(), bd["rois"].cpu().numpy(), atol=1e-4)
    np.testing.assert_array_equal(it["roi_labels"].cpu().numpy(), bd["roi_labels"].cpu().numpy())
    np.testing.assert_allclose(it["batch_box_preds"].cpu().numpy(), bd["batch_box_preds"].cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(it["batch_cls_preds"].cpu().numpy(), bd["batch_cls_preds"].cpu().numpy(), atol=2e-4)
    # second stage in isolation: the MODULE head (torch Linear / BatchNorm1d stacks, dense voxel2pinds volume) on the engine's own RoIs
    # and levels predicts what the engine's folded GEMMs + index queries predict
    bd2 = dict(batch_size=3, rois=it["rois"], roi_labels=it["roi_labels"], roi_scores=it["roi_scores"], has_class_labels=True,
               multi_scale_3d_features=it["levels"], multi_scale_3d_strides={"x_conv3": 4, "x_conv4": 8})
    with torch.no_grad():
        bd2 = net.roi_head(bd2)
    np.testing.assert_allclose(it["batch_box_preds"].cpu().numpy(), bd2["batch_box_preds"].cpu().numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(it["batch_cls_preds"].cpu().numpy(), bd2["batch_cls_preds"].cpu().numpy(), atol=2e-4, rtol=1e-4)
    # post_processing in isolation: the module's per-frame post_processing on the engine's own predictions == the engine's batched one
    bd3 = dict(batch_size=3, batch_box_preds=it["batch_box_preds"], batch_cls_preds=it["batch_cls_preds"], cls_preds_normalized=False,
               has_class_labels=True, roi_labels=it["roi_labels"])
    iso, _ = net.post_processing(bd3)
    for b in range(3):
        _match(got[b], iso[b], 1e-6)
        assert torch.equal(got[b]["pred_boxes"], iso[b]["pred_boxes"]) and torch.equal(got[b]["pred_labels"], iso[b]["pred_labels"])
    # end to end against the module composition: its first stage differs by fp32 rounding (other kernels, other row order), and this
    # random-weight model scores ~20 RoIs that pool nothing within 1e-7 of each other -- their rank, hence which of them the final
    # NMS keeps, is decided by that rounding. Everything else must agree.
    def loose(g, w, atol):
        assert abs(len(g["pred_boxes"]) - len(w["pred_boxes"])) <= 2
        x, y = g["pred_boxes"].cpu().numpy(), w["pred_boxes"].cpu().numpy()
        d = np.abs(x[:, None, :] - y[None, :, :]).max(-1).min(1)
        assert (d <= atol).mean() >= 0.75, float((d <= atol).mean())     # (the near-tied group is ~6 of ~35 boxes)
    for b in range(3):
        loose(got[b], want[b], 1e-3)
    for g, w in zip(eng.forward(clouds), want):           # the default arithmetic (f16x2 first stage)
        loose(g, w, 2e-3)


def test_post_processing_matches_a_plain_restatement(hip):
    """detector3d_template.py:222-343 with MULTI_CLASSES_NMS False, has_class_labels True: sigmoid, score threshold, top-k order,
    rotated NMS at 0.3, labels from roi_labels -- restated with torch + the B3 IoU operator, greedy loop on the host."""
    from cpd_amd import iou3d_nms_utils as iu
    net = _model(seed=5)
    g = torch.Generator().manual_seed(2)
    B, R = 2, 120
    boxes = torch.cat([torch.rand(B, R, 2, generator=g) * 30 - 15, torch.rand(B, R, 1, generator=g), torch.rand(B, R, 3, generator=g) * 3 + 1,
                       torch.rand(B, R, 1, generator=g) * 6 - 3], dim=-1).cuda()
    cls = (torch.randn(B, R, 1, generator=g) * 2).cuda()
    cls[:, -7:] = -9.0                                                        # below SCORE_THRESH after the sigmoid
    labels = torch.randint(1, 4, (B, R), generator=g).cuda()
    bd = dict(batch_size=B, batch_box_preds=boxes, batch_cls_preds=cls, cls_preds_normalized=False, has_class_labels=True, roi_labels=labels)
    got, _ = net.post_processing(bd)
    for b in range(B):
        s = torch.sigmoid(cls[b, :, 0])
        idx = torch.nonzero(s >= 0.01).view(-1)
        order = idx[torch.argsort(s[idx], descending=True)]
        iou = iu.boxes_iou_bev(boxes[b][order], boxes[b][order]).cpu().numpy()
        keep, dead = [], np.zeros(len(order), bool)
        for i in range(len(order)):
            if dead[i]:
                continue
            keep.append(i)
            dead |= iou[i] > 0.3
        sel = order[torch.tensor(keep, device=order.device)]
        assert len(got[b]["pred_boxes"]) == len(sel) < R - 7
        torch.testing.assert_close(got[b]["pred_boxes"], boxes[b][sel])
        torch.testing.assert_close(got[b]["pred_scores"], s[sel])
        assert torch.equal(got[b]["pred_labels"], labels[b][sel])
