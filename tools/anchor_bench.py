#!/usr/bin/env python3
"""Anchor head row at Waymo BEV size: 188x188x2 anchors per class, 3 classes, B frames."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import anchor_head as ah

B, M = 8, 60
cfgs = [dict(class_name=n, anchor_sizes=[s], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0], matched_threshold=0.55,
             unmatched_threshold=0.4) for n, s in (("Vehicle", [4.7, 2.1, 1.7]), ("Pedestrian", [0.91, 0.86, 1.73]), ("Cyclist", [1.78, 0.84, 1.78]))]
anchors, _ = ah.AnchorGenerator([-75.2, -75.2, -2, 75.2, 75.2, 4], cfgs).generate_anchors([[188, 188]] * 3)
rng = np.random.default_rng(0)
sizes = np.array([[4.7, 2.1, 1.7], [0.91, 0.86, 1.73], [1.78, 0.84, 1.78]])
gt = np.zeros((B, M, 8), np.float32)
for b in range(B):
    c = rng.integers(1, 4, M)
    gt[b] = np.concatenate([rng.uniform(-70, 70, (M, 2)), rng.uniform(-1, 1, (M, 1)), sizes[c - 1] * rng.uniform(0.85, 1.15, (M, 3)),
                            rng.uniform(-3.1, 3.1, (M, 1)), c[:, None]], 1)
gt = torch.from_numpy(gt).cuda()
assigner = ah.AxisAlignedTargetAssigner(cfgs, ["Vehicle", "Pedestrian", "Cyclist"])


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


n_anchor = sum(a.numel() // 7 for a in anchors)
t_assign = timed(lambda: assigner.assign_targets(anchors, gt))
a0 = anchors[0].view(-1, 7)
g0 = gt[0, :, :7].contiguous()
c0 = torch.ones(M, dtype=torch.int32, device="cuda")
t_single = timed(lambda: ah.assign_targets_single(a0, g0, c0, 0.55, 0.4), 20)
t_matrix = timed(lambda: ah.boxes3d_nearest_bev_iou(a0, g0), 20)
box = torch.randn(B, 188, 188, 6 * 7, device="cuda") * 0.3
cls = torch.randn(B, 188, 188, 18, device="cuda")
dr = torch.randn(B, 188, 188, 12, device="cuda")
t_dec = timed(lambda: ah.generate_predicted_boxes(anchors, B, cls, box, dr), 20)
print("assign_targets B=%d, %d anchors x %d GT per frame: %.2f ms (host loop over frames/classes); one class of one frame "
      "(fused, no IoU matrix): %.1f us = %.1f G IoU/s; IoU matrix alone %.1f us" %
      (B, n_anchor, M, t_assign, t_single * 1e3, 2 * a0.shape[0] * M / (t_single * 1e-3) / 1e9, t_matrix * 1e3))
print("generate_predicted_boxes B=%d x %d anchors: %.1f us (%.0f GB/s)" % (B, n_anchor, t_dec * 1e3, B * n_anchor * (7 + 7 + 2) * 4 / (t_dec * 1e-3) / 1e9))
tg = assigner.assign_targets(anchors, gt)
lab, reg = tg["box_cls_labels"], tg["box_reg_targets"]
t_loss = timed(lambda: ah.anchor_head_loss(anchors, cls, box, dr, lab, reg, 3), 20)
leaf = [t.clone().requires_grad_(True) for t in (cls, box, dr)]


def torch_loss():
    for t in leaf:
        t.grad = None
    ah.anchor_head_loss_torch(anchors, leaf[0], leaf[1], leaf[2], lab, reg, 3)[0].backward()


t_torch = timed(torch_loss, 5)
nbytes = B * n_anchor * ((3 + 7 + 2) * 2 + 7 + 1) * 4
print("get_loss + gradient B=%d x %d anchors: fused kernel %.1f us (%.0f GB/s algorithmic); torch autograd restatement on the same GPU %.2f ms"
      % (B, n_anchor, t_loss * 1e3, nbytes / (t_loss * 1e-3) / 1e9, t_torch))
