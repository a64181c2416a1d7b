"""Anchor head: the oracle against goldens produced by the reference's own torch code (CPU), the
HIP kernels against the oracle and the same goldens (GPU). Labels are exact; floats <= 1e-6 (same fp32 op order)."""
import numpy as np
import pytest

CLASSES = ["Vehicle", "Pedestrian", "Cyclist"]
THR = {"Vehicle": (0.55, 0.4), "Pedestrian": (0.5, 0.35), "Cyclist": (0.5, 0.35)}


def _assemble(per_class, hw):
    """reference layout: per class (1, H, W*2[, 7]) concatenated on the last (resp. -2) axis, then flattened"""
    H, W = hw
    lab = np.concatenate([t[0].reshape(1, H, -1) for t in per_class], -1).reshape(-1)
    tgt = np.concatenate([t[1].reshape(1, H, -1, 7) for t in per_class], -2).reshape(-1, 7)
    w = np.concatenate([t[2].reshape(1, H, -1) for t in per_class], -1).reshape(-1)
    iou = np.concatenate([t[3].reshape(1, H, -1) for t in per_class], -1).reshape(-1)
    return lab, tgt, w, iou


def test_oracle_reproduces_reference_assigner_and_coders(oracle, golden):
    g = golden("anchor_head")
    hw = [int(v) for v in g["hw"]]
    apc = g["anchors_per_class"]                                     # (3, 1, H, W, 1, 2, 7)
    np.testing.assert_array_equal(oracle.nearest_bev_iou(apc[0].reshape(-1, 7)[::37], g["gt"][0, :14, :7]), g["iou_sample"])
    np.testing.assert_allclose(oracle.residual_encode(g["gt"][0, :14, :7], apc[0].reshape(-1, 7)[:14]), g["enc"], rtol=0, atol=1e-6)
    for b in range(2):
        gt = g["gt"][b]
        gt = gt[:np.nonzero(np.abs(gt).sum(1))[0][-1] + 1]
        per_class = []
        for ci, name in enumerate(CLASSES):
            sel = gt[gt[:, 7].astype(int) == ci + 1]
            per_class.append(oracle.anchor_assign(apc[ci].reshape(-1, 7), sel[:, :7], sel[:, 7].astype(np.int32), *THR[name]))
        lab, tgt, w, iou = _assemble(per_class, hw)
        np.testing.assert_array_equal(lab, g["labels"][b])
        np.testing.assert_array_equal(iou, g["gt_ious"][b])
        np.testing.assert_array_equal(w, g["reg_weights"][b])
        np.testing.assert_allclose(tgt, g["reg_targets"][b], rtol=0, atol=1e-6)
    anchors = np.concatenate(list(apc), axis=-3).reshape(-1, 7)       # torch.cat(self.anchors, dim=-3), l.354
    dec = oracle.anchor_decode(g["box_preds"].reshape(2, -1, 7), anchors, g["dir_preds"].reshape(2, -1, 2))
    np.testing.assert_allclose(dec, g["decoded"], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_hip_anchor_head_matches_reference_goldens(oracle, golden, hip):
    import torch
    from cpd_amd import anchor_head as ah
    g = golden("anchor_head")
    H, W = [int(v) for v in g["hw"]]
    cfgs = [dict(class_name="Vehicle", anchor_sizes=[[4.7, 2.1, 1.7]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0],
                 matched_threshold=0.55, unmatched_threshold=0.4),
            dict(class_name="Pedestrian", anchor_sizes=[[0.91, 0.86, 1.73]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0],
                 matched_threshold=0.5, unmatched_threshold=0.35),
            dict(class_name="Cyclist", anchor_sizes=[[1.78, 0.84, 1.78]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0],
                 matched_threshold=0.5, unmatched_threshold=0.35)]
    anchors, per_loc = ah.AnchorGenerator(g["pcr"].tolist(), cfgs).generate_anchors([[W, H]] * 3)
    assert per_loc == [2, 2, 2]
    np.testing.assert_allclose(torch.stack(anchors).cpu().numpy(), g["anchors_per_class"], rtol=0, atol=1e-5)
    anchors = [torch.from_numpy(a).cuda() for a in g["anchors_per_class"]]          # the reference's exact anchors from here on
    iou = ah.boxes3d_nearest_bev_iou(anchors[0].view(-1, 7)[::37], torch.from_numpy(g["gt"][0, :14, :7]).cuda())
    np.testing.assert_array_equal(iou.cpu().numpy(), g["iou_sample"])
    tgt = ah.AxisAlignedTargetAssigner(cfgs, CLASSES).assign_targets(anchors, torch.from_numpy(g["gt"]).cuda())
    np.testing.assert_array_equal(tgt["box_cls_labels"].cpu().numpy(), g["labels"])
    np.testing.assert_array_equal(tgt["gt_ious"].cpu().numpy(), g["gt_ious"])
    np.testing.assert_array_equal(tgt["reg_weights"].cpu().numpy(), g["reg_weights"])
    np.testing.assert_allclose(tgt["box_reg_targets"].cpu().numpy(), g["reg_targets"], rtol=0, atol=2e-6)
    cls, dec = ah.generate_predicted_boxes(anchors, 2, torch.from_numpy(g["cls_preds"]).cuda(), torch.from_numpy(g["box_preds"]).cuda(),
                                           torch.from_numpy(g["dir_preds"]).cuda())
    np.testing.assert_allclose(dec.cpu().numpy(), g["decoded"], rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(cls.cpu().numpy(), g["batch_cls"])
    # a Waymo-size assignment (212k anchors per class x 60 GT) against the oracle, incl. a class without GT
    rng = np.random.default_rng(1)
    big = ah.AnchorGenerator([-75.2, -75.2, -2, 75.2, 75.2, 4], cfgs[:1]).generate_anchors([[188, 188]])[0][0].view(-1, 7)
    gt = np.concatenate([rng.uniform(-70, 70, (60, 2)), rng.uniform(-1, 1, (60, 1)), np.array([4.7, 2.1, 1.7]) * rng.uniform(0.8, 1.2, (60, 3)),
                         rng.uniform(-3.1, 3.1, (60, 1))], 1).astype(np.float32)
    want = oracle.anchor_assign(big.cpu().numpy(), gt, np.ones(60, np.int32), 0.55, 0.4)
    got = ah.assign_targets_single(big, torch.from_numpy(gt).cuda(), torch.ones(60, dtype=torch.int32).cuda(), 0.55, 0.4)
    np.testing.assert_array_equal(got["box_cls_labels"].cpu().numpy(), want[0])
    np.testing.assert_array_equal(got["gt_ious"].cpu().numpy(), want[3])
    np.testing.assert_allclose(got["box_reg_targets"].cpu().numpy(), want[1], rtol=0, atol=2e-6)
    none = ah.assign_targets_single(big, torch.zeros((0, 7)).cuda(), torch.zeros((0,), dtype=torch.int32).cuda(), 0.55, 0.4)
    assert int(none["box_cls_labels"].abs().sum()) == 0 and float(none["reg_weights"].sum()) == 0.0


@pytest.mark.gpu
def test_anchor_head_single_forward_matches_reference_module(golden, hip):
    """The whole eval forward of the reference's AnchorHeadSingle (mask, 1x1 convs, masked anchors, decoding)."""
    import torch
    from cpd_amd import anchor_head as ah
    g = golden("anchor_head_single")
    cfgs = [dict(class_name=n, anchor_sizes=[s], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0], align_center=False,
                 feature_map_stride=8, matched_threshold=0.5, unmatched_threshold=0.35)
            for n, s in (("Vehicle", [4.7, 2.1, 1.7]), ("Pedestrian", [0.91, 0.86, 1.73]), ("Cyclist", [1.78, 0.84, 1.78]))]
    mcfg = dict(ANCHOR_GENERATOR_CONFIG=cfgs, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2)
    head = ah.AnchorHeadSingle(mcfg, 24, 3, CLASSES, np.array([416, 416, 40]), g["pcr"].tolist())
    sd = {k[4:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("ahs.")}
    missing = head.load_state_dict(sd, strict=False)
    assert not missing.missing_keys                                  # every parameter of ours exists in the reference module
    head = head.cuda().eval()
    dd = {"points": torch.from_numpy(g["points"]).cuda(), "st_features_2d": torch.from_numpy(g["feat"]).cuda(), "batch_size": 2}
    mask = head.get_anchor_mask(dd["points"], dd["st_features_2d"].shape)
    np.testing.assert_array_equal(mask.cpu().numpy(), g["mask"])
    out = head(dd)
    np.testing.assert_allclose(out["batch_cls_preds"].cpu().numpy(), g["batch_cls_preds"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["batch_box_preds"].cpu().numpy(), g["batch_box_preds"], atol=2e-4, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("math", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_anchor_head_single_v2_forward_matches_reference_module(golden, hip, math, mode):
    """AnchorHeadSingleV2 (anchor_head_single.py:31-192), the head the shipped dbscan / oyster configs select: state_dict of the
    reference module loads by name, same anchor mask, and the raw cls / box / dir maps plus the decoded boxes equal the
    reference's own eval forward -- through the fused four-launch eval path (every conv arithmetic) and through the
    differentiable module-by-module path train mode takes (BatchNorm put in eval so that both see the running statistics)."""
    import torch
    from cpd_amd import anchor_head as ah
    g = golden("anchor_head_single_v2")
    cfgs = [dict(class_name=n, anchor_sizes=[s], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0], align_center=False,
                 feature_map_stride=8, matched_threshold=0.5, unmatched_threshold=0.35)
            for n, s in (("Vehicle", [4.7, 2.1, 1.7]), ("Pedestrian", [0.91, 0.86, 1.73]), ("Cyclist", [1.78, 0.84, 1.78]))]
    mcfg = dict(ANCHOR_GENERATOR_CONFIG=cfgs, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2,
                TARGET_ASSIGNER_CONFIG=dict(NAME="AxisAlignedTargetAssigner", NORM_BY_NUM_EXAMPLES=False, MATCH_HEIGHT=False))
    head = ah.AnchorHeadSingleV2(mcfg, 32, 3, CLASSES, np.array([416, 416, 40]), g["pcr"].tolist(), conv_math=math)
    sd = {k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("v2.")}
    res = head.load_state_dict(sd, strict=True)                      # same parameter / buffer names as the reference class
    assert not res.missing_keys and not res.unexpected_keys
    head = head.cuda().eval()
    if mode == "train":
        head.training = True                                         # forward takes the module path; BatchNorms stay in eval
    dd = {"points": torch.from_numpy(g["points"]).cuda(), "st_features_2d": torch.from_numpy(g["feat"]).cuda(), "batch_size": 2}
    if mode == "train":                                              # training mode assigns targets (anchor_head_single.py:176-181)
        dd["gt_boxes"] = torch.tensor([[[3.0, -2.0, 0.0, 4.5, 2.0, 1.6, 0.3, 1.0], [-6.0, 5.0, 0.1, 0.9, 0.8, 1.7, 1.2, 2.0]]] * 2).cuda()
    mask = head.get_anchor_mask(dd["points"], dd["st_features_2d"].shape)
    np.testing.assert_array_equal(mask.cpu().numpy(), g["mask"])
    out = head(dd)
    f = head.forward_ret_dict
    np.testing.assert_allclose(f["cls_preds"].detach().cpu().numpy(), g["cls_preds"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(f["box_preds"].detach().cpu().numpy(), g["box_preds"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(f["dir_cls_preds"].detach().cpu().numpy(), g["dir_cls_preds"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["batch_cls_preds"].cpu().numpy(), g["batch_cls_preds"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["batch_box_preds"].cpu().numpy(), g["batch_box_preds"], atol=5e-4, rtol=1e-4)
    if mode == "train":                                              # the module path is differentiable end to end
        n_anchor = g["batch_box_preds"].shape[1]
        assert f["box_cls_labels"].shape == (2, n_anchor) and f["box_reg_targets"].shape == (2, n_anchor, 7)
        assert out["gt_ious"].shape == (2, n_anchor)
        (f["cls_preds"].sum() + f["box_preds"].sum()).backward()
        assert all(p.grad is not None for n, p in head.named_parameters() if not n.startswith("conv_dir_cls"))


@pytest.mark.gpu
def test_v2_head_and_proto_head_train_one_step_end_to_end(golden, hip):
    """with the shipped voxel_rcnn configs a training step runs AnchorHeadSingleV2 (targets on the occupancy-masked
    anchors, boxes predicted because predict_boxes_when_training defaults to True as in the reference ctor) and hands
    batch_box_preds / batch_cls_preds to VoxelRCNNProtoHead.proposal_layer. One such step, end to end: V2's fused get_loss on
    its own forward_ret_dict (default = masked anchors) equals the autograd restatement (loss and gradients), its gradients reach
    the V2 parameters, and the second stage trains on V2's proposals."""
    import types
    import torch
    from cpd_amd import anchor_head as ah
    from cpd_amd.roi_head_train import VoxelRCNNProtoHead
    from test_gpu_roi_train import _cfg
    g2, gp = golden("anchor_head_single_v2"), golden("proto_head")
    cfgs = [dict(class_name=n, anchor_sizes=[s], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0], align_center=False,
                 feature_map_stride=8, matched_threshold=0.5, unmatched_threshold=0.35)
            for n, s in (("Vehicle", [4.7, 2.1, 1.7]), ("Pedestrian", [0.91, 0.86, 1.73]), ("Cyclist", [1.78, 0.84, 1.78]))]
    mcfg = dict(ANCHOR_GENERATOR_CONFIG=cfgs, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2,
                TARGET_ASSIGNER_CONFIG=dict(NAME="AxisAlignedTargetAssigner", NORM_BY_NUM_EXAMPLES=False, MATCH_HEIGHT=False),
                LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=[1.0] * 7)))
    head = ah.AnchorHeadSingleV2(mcfg, 32, 3, CLASSES, np.array([416, 416, 40]), g2["pcr"].tolist(), conv_math="f32")
    head.load_state_dict({k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g2.items() if k.startswith("v2.")})
    head = head.cuda().train()                                      # REAL training mode: batch-statistics BatchNorm
    gt = torch.from_numpy(gp["gt"]).cuda()
    dd = {"points": torch.from_numpy(g2["points"]).cuda(), "st_features_2d": torch.from_numpy(g2["feat"]).cuda().requires_grad_(True),
          "batch_size": 2, "gt_boxes": gt}
    dd = head(dd)
    f = head.forward_ret_dict
    assert "batch_box_preds" in dd and "gt_ious" in dd and "box_cls_labels" in f and "box_reg_targets" in f
    assert int((f["box_cls_labels"] > 0).sum()) > 0                  # the scene has matched anchors
    losses, grads = head.get_loss()                                  # defaults: own forward_ret_dict, masked anchors
    det = {k: (v.detach().requires_grad_(True) if k.endswith("preds") and v is not None else v) for k, v in f.items()}
    want, _ = ah.anchor_head_loss_torch(head.anchors, det["cls_preds"], det["box_preds"], det["dir_cls_preds"], f["box_cls_labels"],
                                        f["box_reg_targets"], 3, 1.0, 2.0, 0.2, [1.0] * 7)
    want.backward()
    assert abs(float(losses[0]) - float(want.detach())) <= 1e-4 * max(1.0, abs(float(want.detach())))
    for got, key in zip(grads, ("cls_preds", "box_preds", "dir_cls_preds")):
        ref = det[key].grad
        assert float((got - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    torch.autograd.backward([f["cls_preds"], f["box_preds"], f["dir_cls_preds"]], list(grads))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in head.parameters())
    assert dd["st_features_2d"].grad is not None
    # second stage on V2's proposals
    roi = VoxelRCNNProtoHead(input_channels={"x_conv3": 8, "x_conv4": 12}, model_cfg=_cfg(), point_cloud_range=gp["pcr"].tolist(),
                             voxel_size=[0.1, 0.1, 0.15], num_class=1).cuda().train()
    lv, lv_mm = {}, {}
    for name, shp in (("x_conv3", [11, 104, 104]), ("x_conv4", [5, 52, 52])):
        idx = torch.from_numpy(gp[name + "_idx"]).cuda()
        for d, key in ((lv, "_feat"), (lv_mm, "_feat_mm")):
            d[name] = types.SimpleNamespace(indices=idx, features=torch.from_numpy(gp[name + key]).cuda().requires_grad_(True),
                                            spatial_shape=shp, batch_size=2)
    dd.update(css_score=torch.from_numpy(gp["css"]).cuda(), multi_scale_3d_features=lv, multi_scale_3d_features_mm=lv_mm,
              multi_scale_3d_strides={"x_conv3": 4, "x_conv4": 8})
    np.random.seed(1); torch.manual_seed(1)
    roi(dd)
    loss, tb = roi.get_loss()
    assert torch.isfinite(loss)
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in roi.parameters() if p.grad is not None)
    assert sum(p.grad is not None for p in roi.parameters()) >= 30


def test_anchor_loss_restatement_matches_reference_get_loss(golden):
    """cpd_amd.anchor_head.anchor_head_loss_torch against the reference's own AnchorHeadTemplate.get_loss and the gradients its
    autograd produced (tests/golden/anchor_loss.npz, section 12 of make_golden.py). CPU."""
    import torch
    from cpd_amd import anchor_head as ah
    a, g = golden("anchor_head"), golden("anchor_loss")
    preds = [torch.tensor(a[k], requires_grad=True) for k in ("cls_preds", "box_preds", "dir_preds")]
    w = g["weights"]
    total, parts = ah.anchor_head_loss_torch(torch.tensor(a["anchors"]), preds[0], preds[1], preds[2], torch.tensor(a["labels"]),
                                             torch.tensor(a["reg_targets"]), 3, float(w[0]), float(w[1]), float(w[2]),
                                             dir_offset=float(g["dir_offset"]))
    total.backward()
    np.testing.assert_allclose(float(total.detach()), g["rpn_loss"], rtol=1e-6)
    for k, name in (("rpn_loss_cls", "cls_loss"), ("rpn_loss_loc", "loc_loss"), ("rpn_loss_dir", "dir_loss")):
        np.testing.assert_allclose(float(parts[k]), g[name], rtol=1e-6)
    n = a["labels"].shape[1]
    for t, name, row in zip(preds, ("g_cls", "g_box", "g_dir"), range(3)):
        full = t.grad.reshape(2, n, -1)
        np.testing.assert_allclose(full[:, ::5].numpy(), g[name], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose([full.sum().item(), full.abs().sum().item()], g["g_sums"][row], rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_fused_anchor_loss_matches_reference_get_loss(golden, hip):
    """cpd_anchor_loss (three launches, loss + gradient) against the same reference goldens, and against the torch
    restatement on a larger random problem with ignored anchors, empty samples and a NaN regression target."""
    import torch
    from cpd_amd import anchor_head as ah
    a, g = golden("anchor_head"), golden("anchor_loss")
    dev = lambda k: torch.tensor(a[k]).cuda()
    w = g["weights"]
    losses, grads = ah.anchor_head_loss(dev("anchors"), dev("cls_preds"), dev("box_preds"), dev("dir_preds"), dev("labels"),
                                        dev("reg_targets"), 3, float(w[0]), float(w[1]), float(w[2]), dir_offset=float(g["dir_offset"]))
    got = losses.cpu().numpy()
    np.testing.assert_allclose(got, [g["rpn_loss"], g["cls_loss"], g["loc_loss"], g["dir_loss"]], rtol=3e-6)
    n = a["labels"].shape[1]
    for t, name, row in zip(grads, ("g_cls", "g_box", "g_dir"), range(3)):
        assert t.shape == tuple(a[{"g_cls": "cls_preds", "g_box": "box_preds", "g_dir": "dir_preds"}[name]].shape)
        full = t.reshape(2, n, -1)
        ref = g[name]
        assert np.abs(full[:, ::5].cpu().numpy() - ref).max() <= 3e-6 * np.abs(ref).max()
        np.testing.assert_allclose([full.sum().item(), full.abs().sum().item()], g["g_sums"][row], rtol=1e-4, atol=1e-6)
    # larger random problem
    torch.manual_seed(5)
    B, n, nc = 3, 40000, 3
    anchors = torch.randn(n, 7, device="cuda")
    cls, box, dr = torch.randn(B, n, nc, device="cuda") * 2, torch.randn(B, n, 7, device="cuda"), torch.randn(B, n, 2, device="cuda")
    labels = torch.randint(-1, nc + 1, (B, n), device="cuda", dtype=torch.int32)
    labels[1] = torch.where(labels[1] > 0, torch.zeros_like(labels[1]), labels[1])          # a sample without positives
    reg = torch.randn(B, n, 7, device="cuda") * 0.5
    reg[0, 7, 2] = float("nan")
    cw = [1.0, 1.0, 1.0, 0.5, 0.5, 0.5, 2.0]
    leaf = [t.clone().requires_grad_(True) for t in (cls, box, dr)]
    want, parts = ah.anchor_head_loss_torch(anchors, leaf[0], leaf[1], leaf[2], labels, reg, nc, 1.0, 2.0, 0.2, cw)
    want.backward()
    losses, grads = ah.anchor_head_loss(anchors, cls, box, dr, labels, reg, nc, 1.0, 2.0, 0.2, cw)
    got = losses.cpu().numpy()
    np.testing.assert_allclose(got, [float(want.detach()), float(parts["rpn_loss_cls"]), float(parts["rpn_loss_loc"]),
                                     float(parts["rpn_loss_dir"])], rtol=1e-5)
    for t, l in zip(grads, leaf):
        assert (t - l.grad).abs().max().item() <= 1e-5 * l.grad.abs().max().item()
    again, _ = ah.anchor_head_loss(anchors, cls, box, dr, labels, reg, nc, 1.0, 2.0, 0.2, cw)
    assert torch.equal(again, losses)


def _trim(gt):
    nz = np.nonzero(gt[:, :7].sum(1))[0]
    return gt[:(nz[-1] + 1 if len(nz) else 1)]


@pytest.mark.parametrize("k", [8, 9])
def test_oracle_atss_reproduces_reference_assigner(oracle, golden, k):
    """atss.npz = the reference's ATSSTargetAssigner itself (CPU torch, reference CPU IoU): k = 8 on two-rotation anchors,
    k = 9 on one-rotation anchors (an odd k on tied same-centre pairs is unspecified in torch.topk: make_golden.py::atss)."""
    g = golden("atss")
    apc = g["anchors_k%d" % k]
    for b in range(2):
        gt = _trim(g["gt"][b])
        lab, tgt, w = zip(*[oracle.atss_assign(apc[c].reshape(-1, 7), gt, k) for c in range(3)])
        np.testing.assert_array_equal(np.concatenate(lab), g["labels_k%d" % k][b])
        np.testing.assert_array_equal(np.concatenate(w), g["reg_weights_k%d" % k][b])
        np.testing.assert_allclose(np.concatenate(tgt), g["reg_targets_k%d" % k][b], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(lab[0], g["labels_single_k%d" % k][b])


@pytest.mark.gpu
@pytest.mark.parametrize("k", [8, 9])
def test_hip_atss_matches_reference_golden(golden, hip, k):
    import torch
    from cpd_amd import anchor_head as ah
    g = golden("atss")
    anchors = [torch.from_numpy(a).cuda() for a in g["anchors_k%d" % k]]
    asg = ah.ATSSTargetAssigner(topk=k, match_height=False)
    t = asg.assign_targets(anchors, torch.from_numpy(g["gt"]).cuda())
    np.testing.assert_array_equal(t["box_cls_labels"].cpu().numpy(), g["labels_k%d" % k])
    np.testing.assert_array_equal(t["reg_weights"].cpu().numpy(), g["reg_weights_k%d" % k])
    np.testing.assert_allclose(t["box_reg_targets"].cpu().numpy(), g["reg_targets_k%d" % k], rtol=0, atol=1e-6)
    t1 = asg.assign_targets(anchors[0], torch.from_numpy(g["gt"]).cuda())
    np.testing.assert_array_equal(t1["box_cls_labels"].cpu().numpy(), g["labels_single_k%d" % k])


@pytest.mark.gpu
@pytest.mark.parametrize("match_height", [False, True])
def test_hip_atss_matches_oracle_at_head_size(oracle, hip, match_height):
    """One class of a 188 x 188 head (70,688 anchors) against 40 boxes, incl. a box no anchor overlaps, near-duplicate boxes
    that compete for the same anchors and an all-zero frame (the reference keeps at least one row): the device path never
    builds the 2.8 M-entry matrices the oracle (and the reference) work on."""
    import torch
    from cpd_amd import anchor_head as ah
    rng = np.random.default_rng(7 + int(match_height))
    cfg = [dict(class_name="Vehicle", anchor_sizes=[[4.7, 2.1, 1.7]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0])]
    pcr = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    anchors, _ = ah.AnchorGenerator(pcr, cfg).generate_anchors([[188, 188]])
    a = anchors[0].reshape(-1, 7)
    m = 40
    gt = np.zeros((1, m + 2, 8), np.float32)
    for i in range(m):
        gt[0, i] = [rng.uniform(-70, 70), rng.uniform(-70, 70), rng.uniform(-1.5, 0.5), *(np.array([4.7, 2.1, 1.7]) * rng.uniform(0.7, 1.3, 3)),
                    rng.uniform(-3.1, 3.1), rng.integers(1, 4)]
    gt[0, 5, :3] = [300.0, 300.0, 0.0]                                 # outside the anchor grid: IoU 0 with everything
    gt[0, 8, :7] = gt[0, 7, :7]
    gt[0, 8, 0] += 0.05                                                # near-duplicate boxes
    t = ah.ATSSTargetAssigner(topk=9, match_height=match_height).assign_targets(anchors[0], torch.from_numpy(gt).cuda())
    lab, tgt, w = oracle.atss_assign(a.cpu().numpy(), gt[0, :m], 9, match_height)
    got = t["box_cls_labels"][0].cpu().numpy()
    assert (got > 0).sum() >= m - 2
    np.testing.assert_array_equal(got, lab)
    np.testing.assert_array_equal(t["reg_weights"][0].cpu().numpy(), w)
    np.testing.assert_allclose(t["box_reg_targets"][0].cpu().numpy(), tgt, rtol=0, atol=1e-5)
    zero = np.zeros((1, 3, 8), np.float32)                             # no boxes at all: one all-zero row survives the trim
    t0 = ah.ATSSTargetAssigner(topk=9).assign_targets(anchors[0], torch.from_numpy(zero).cuda())
    assert float(t0["box_cls_labels"].abs().sum()) == 0 and float(t0["reg_weights"].sum()) == 0
