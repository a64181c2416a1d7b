"""Second-stage TRAINING branch (VoxelRCNNProtoHead) against tests/golden/proto_head.npz, which the reference's own classes produced
on CPU (make_golden.py::proto_head: proposal layer, proposal-target sampling under recorded seeds, canonical targets, both pooling
branches with batch-statistics BatchNorm, get_loss, autograd gradients). Sampled indices / labels / masks are exact (the sampler
draws the same random numbers); floats agree to fp32 accumulation-order noise (BatchNorm over thousands of rows, 1e-4 relative)."""
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg():
    def pool():
        return dict(FEATURES_SOURCE=["x_conv3", "x_conv4"], PRE_MLP=True, GRID_SIZE=2, POOL_LAYERS=dict(
            x_conv3=dict(MLPS=[[16, 16], [16, 16]], QUERY_RANGES=[[1, 1, 1], [2, 2, 2]], POOL_RADIUS=[0.6, 1.2], NSAMPLE=[8, 8], POOL_METHOD="max_pool"),
            x_conv4=dict(MLPS=[[16, 16], [16, 16]], QUERY_RANGES=[[1, 1, 1], [2, 2, 2]], POOL_RADIUS=[1.2, 2.4], NSAMPLE=[8, 8], POOL_METHOD="max_pool")))
    return dict(
        CLASS_AGNOSTIC=True, ROI_GRID_POOL=pool(), ROI_GRID_POOL_PROTO=pool(), SHARED_FC=[48, 48], CLS_FC=[32, 32], REG_FC=[32, 32], DP_RATIO=0.0,
        TARGET_CONFIG=dict(BOX_CODER="ResidualCoder", ROI_PER_IMAGE=24, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE="roi_iou",
                           CLS_FG_THRESH=0.6, CLS_BG_THRESH=0.02, CLS_BG_THRESH_LO=0.01, HARD_BG_RATIO=0.1, REG_FG_THRESH=0.3),
        LOSS_CONFIG=dict(CLS_LOSS="BinaryCrossEntropy", REG_LOSS="smooth-l1", CORNER_LOSS_REGULARIZATION=True, GRID_3D_IOU_LOSS=False,
                         LOSS_WEIGHTS=dict(rcnn_proto_weight=1.0, rcnn_cls_weight=1.0, rcnn_reg_weight=1.0, rcnn_corner_weight=1.0, rcnn_iou3d_weight=1.0,
                                           code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.8])),
        NMS_CONFIG=dict(TRAIN=dict(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=400, NMS_POST_MAXSIZE=60, NMS_THRESH=0.8)))


def _close(got, want, what, rtol=2e-4, floor=2e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(float(np.abs(want).max()), 1e-12)
    err = float(np.abs(got - want).max())
    assert err <= rtol * scale + floor, "%s: max |diff| %.3e vs scale %.3e" % (what, err, scale)


def test_proto_head_training_step_matches_reference(golden, hip):
    import torch
    from cpd_amd.roi_head_train import VoxelRCNNProtoHead
    g = golden("proto_head")
    head = VoxelRCNNProtoHead(input_channels={"x_conv3": 8, "x_conv4": 12}, model_cfg=_cfg(), point_cloud_range=g["pcr"].tolist(),
                              voxel_size=[0.1, 0.1, 0.15], num_class=1)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("h.")}
    missing, unexpected = head.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)          # same parameter / buffer names as the reference class
    head = head.cuda().train()
    lv, lv_mm = {}, {}
    for name, shp in (("x_conv3", [11, 104, 104]), ("x_conv4", [5, 52, 52])):
        idx = torch.from_numpy(g[name + "_idx"]).cuda()
        for d, key in ((lv, "_feat"), (lv_mm, "_feat_mm")):
            f = torch.from_numpy(g[name + key]).cuda().requires_grad_(True)
            d[name] = types.SimpleNamespace(indices=idx, features=f, spatial_shape=shp, batch_size=2)
    bd = {"batch_size": 2, "batch_box_preds": torch.from_numpy(g["boxes"]).cuda(), "batch_cls_preds": torch.from_numpy(g["cls"]).cuda(),
          "gt_boxes": torch.from_numpy(g["gt"]).cuda(), "css_score": torch.from_numpy(g["css"]).cuda(), "multi_scale_3d_features": lv,
          "multi_scale_3d_features_mm": lv_mm, "multi_scale_3d_strides": {"x_conv3": 4, "x_conv4": 8}}
    np.random.seed(int(g["seed"]))
    torch.manual_seed(int(g["seed"]))
    from cpd_amd import ops
    from cpd_amd.autograd_ops import HipConv1d, HipLinear
    with ops.launch_log() as fwd_log:
        head(bd)
    # round 4: the FC stacks and the pooling MLPs' 1 x 1 convs are C-ABI launches in TRAINING too (no rocBLAS GEMM on the path):
    # every Linear layer of the six stacks and the 16 Conv1d layers (2 branches x 2 levels x 2 scales x in / out), one launch each
    n_fc = sum(isinstance(m, HipLinear) for m in head.modules())
    n_c1 = sum(isinstance(m, HipConv1d) for m in head.modules())
    assert n_fc >= 12 and n_c1 == 16 and sum(fwd_log.counts.values()) >= n_fc + n_c1, (n_fc, n_c1, fwd_log.counts)
    # round 5: no torch BatchNorm / Conv2d module is left in the head -- BatchNorm1d / 2d and the 3 -> C position convs are the C-ABI's too
    from cpd_amd.autograd_ops import HipBatchNorm1d, HipBatchNorm2d, HipPointwiseConv2d
    for m in head.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            assert isinstance(m, (HipBatchNorm1d, HipBatchNorm2d)), type(m)
        if isinstance(m, torch.nn.Conv2d):
            assert isinstance(m, HipPointwiseConv2d), type(m)
    assert sum(isinstance(m, HipPointwiseConv2d) for m in head.modules()) == 8
    assert all(int(m.num_batches_tracked) == 1 for m in head.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm))
    t0, t1 = head.forward_ret_dict["targets_dict0"], head.forward_ret_dict["targets_dict1"]
    # sampling and targets
    np.testing.assert_allclose(t0["rois"].cpu().numpy(), g["t_rois"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(t0["roi_labels"].cpu().numpy(), g["t_roi_labels"])
    np.testing.assert_array_equal(t0["reg_valid_mask"].cpu().numpy(), g["t_reg_valid_mask"])
    np.testing.assert_allclose(t0["gt_iou_of_rois"].cpu().numpy(), g["t_gt_iou_of_rois"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(t0["rcnn_cls_labels"].cpu().numpy(), g["t_rcnn_cls_labels"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(t0["gt_of_rois"].cpu().numpy(), g["t_gt_of_rois"], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(t0["gt_of_rois_src"].cpu().numpy(), g["t_gt_of_rois_src"])
    np.testing.assert_array_equal(t0["additional_data"]["css_score"].cpu().numpy(), g["t_css"])
    np.testing.assert_allclose(t0["roi_scores"].cpu().numpy(), g["t_roi_scores"], rtol=0, atol=1e-6)
    # both branches' outputs
    for i, t in enumerate((t0, t1)):
        for k in ("shared_features", "rcnn_cls", "rcnn_reg"):
            _close(t[k].detach().cpu().numpy(), g["o%d_%s" % (i, k)], "branch %d %s" % (i, k))
    loss, tb = head.get_loss()
    assert abs(loss.item() - float(g["loss"])) <= 2e-4 * abs(float(g["loss"])), (loss.item(), float(g["loss"]))
    assert abs(tb["rcnn_loss"] - float(g["rcnn_loss"])) <= 2e-4 * abs(float(g["rcnn_loss"]))
    loss.backward()
    # gradients: into the parameters the fixture holds and into all four sparse feature tensors (every 4th row stored)
    params = dict(head.named_parameters())
    n = 0
    for k in g.files:
        if k.startswith("g."):
            _close(params[k[2:]].grad.cpu().numpy(), g[k], "grad " + k[2:], rtol=1e-3, floor=1e-6)
            n += 1
    assert n >= 30
    for name in ("x_conv3", "x_conv4"):
        _close(lv[name].features.grad.cpu().numpy()[::4], g[name + "_grad4"], "d loss / d " + name, rtol=1e-3, floor=1e-7)
        _close(lv_mm[name].features.grad.cpu().numpy()[::4], g[name + "_grad4_mm"], "d loss / d mm " + name, rtol=1e-3, floor=1e-7)
    assert head.iter == 2


def test_grouping_backward_is_the_scatter_add_of_the_forward(hip):
    """GroupingOperation: backward(cpd_group_points_grad) against torch.index_add on the same indices."""
    import torch
    from cpd_amd import roi_pool as rp
    gen = torch.Generator().manual_seed(5)
    n0, n1, m0, m1, c, ns = 700, 900, 300, 450, 24, 16
    feats = torch.randn(n0 + n1, c, generator=gen).cuda().requires_grad_(True)
    fcnt = torch.tensor([n0, n1], dtype=torch.int32).cuda()
    icnt = torch.tensor([m0, m1], dtype=torch.int32).cuda()
    idx = torch.cat([torch.randint(0, n0, (m0, ns), generator=gen), torch.randint(0, n1, (m1, ns), generator=gen)]).int().cuda()
    out = rp.GroupingOperation.apply(feats, fcnt, idx, icnt)
    w = torch.randn(out.shape, generator=gen).cuda()
    (out * w).sum().backward()
    rows = idx.long() + torch.cat([torch.zeros(m0, 1), torch.full((m1, 1), n0)]).long().cuda()
    want = torch.zeros(n0 + n1, c, dtype=torch.float64, device="cuda").index_add_(0, rows.reshape(-1), w.permute(0, 2, 1).reshape(-1, c).double())
    assert torch.equal(out, feats.detach()[rows].permute(0, 2, 1))
    assert float((feats.grad.double() - want).abs().max()) < 1e-5


def test_per_class_threshold_target_layer_matches_reference(golden, hip):
    """ProposalTargetLayer with CLS_SCORE_TYPE roi_iou_x / roi_ioud_x (per-class threshold lists; proposal_target_layer.py:57-80, 128-184,
    258-259, 297-322), with and without ENABLE_HARD_SAMPLING, against tests/golden/target_layer_x.npz -- the reference's own class on CPU
    under recorded seeds (make_golden.py::target_layer_x). The sampler draws the same random numbers in the same order: sampled RoIs,
    labels and the regression mask are exact; IoUs and the soft class labels to fp32 rounding of the device IoU."""
    import torch
    from cpd_amd.roi_head_train import ProposalTargetLayer
    g = golden("target_layer_x")
    bd = {"batch_size": 2, "rois": torch.from_numpy(g["rois"]).cuda(), "roi_scores": torch.from_numpy(g["roi_scores"]).cuda(),
          "roi_labels": torch.from_numpy(g["roi_labels"]).cuda(), "gt_boxes": torch.from_numpy(g["gt"]).cuda()}
    for case in range(int(g["n_cases"])):
        pre = "c%d_" % case
        cfg = dict(ROI_PER_IMAGE=64, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE=("roi_iou_x", "roi_ioud_x")[int(g[pre + "kind"])],
                   CLS_FG_THRESH=[0.75, 0.6, 0.65], CLS_BG_THRESH=[0.25, 0.15, 0.2], CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8,
                   REG_FG_THRESH=[0.55, 0.4, 0.45], DIRECTION_MIN=0.1, DIRECTION_MAX=0.9, ENABLE_HARD_SAMPLING=bool(g[pre + "hard"]),
                   HARD_SAMPLING_THRESH=[0.3, 0.2, 0.25], HARD_SAMPLING_RATIO=[0.5, 0.25, 0.34])
        np.random.seed(int(g[pre + "seed"]))
        torch.manual_seed(int(g[pre + "seed"]))
        t = ProposalTargetLayer(cfg)(dict(bd))
        np.testing.assert_array_equal(t["rois"].cpu().numpy(), g[pre + "rois"], err_msg=pre)
        np.testing.assert_array_equal(t["gt_of_rois"].cpu().numpy(), g[pre + "gt_of_rois"], err_msg=pre)
        np.testing.assert_array_equal(t["roi_labels"].cpu().numpy(), g[pre + "roi_labels"], err_msg=pre)
        np.testing.assert_array_equal(t["roi_scores"].cpu().numpy(), g[pre + "roi_scores"], err_msg=pre)
        np.testing.assert_array_equal(t["reg_valid_mask"].cpu().numpy(), g[pre + "reg_valid_mask"], err_msg=pre)
        np.testing.assert_allclose(t["gt_iou_of_rois"].cpu().numpy(), g[pre + "gt_iou_of_rois"], rtol=0, atol=1e-5, err_msg=pre)
        np.testing.assert_allclose(t["rcnn_cls_labels"].cpu().numpy(), g[pre + "rcnn_cls_labels"], rtol=0, atol=5e-5, err_msg=pre)
        assert int(t["reg_valid_mask"].sum()) > 20 and 0.0 < float(t["rcnn_cls_labels"].mean()) < 1.0
