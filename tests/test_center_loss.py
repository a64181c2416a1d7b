"""cpd_amd.center_loss (vectorised torch) against golden vectors produced by the reference's own
CenterHead.assign_target_of_single_head, FocalLossCenterNet and RegLossCenterNet."""
import numpy as np
import torch

from cpd_amd import center_loss as cl


def test_assign_targets_match_reference(golden):
    g = golden("center_loss")
    gt = torch.from_numpy(g["gt_boxes"])[None]
    heat, tgt, inds, masks = cl.assign_targets(gt, (188, 188), [-75.2, -75.2, -2, 75.2, 75.2, 4], [0.1, 0.1, 0.15], 3,
                                               feature_map_stride=8, num_max_objs=500, gaussian_overlap=0.1, min_radius=2)
    np.testing.assert_allclose(heat[0].numpy(), g["heatmap"], atol=1e-6)
    np.testing.assert_array_equal(masks[0].numpy(), g["mask"])
    np.testing.assert_array_equal(inds[0].numpy(), g["inds"])
    np.testing.assert_allclose(tgt[0].numpy(), g["ret_boxes"], atol=1e-5)


def test_losses_match_reference(golden):
    g = golden("center_loss")
    focal = cl.neg_loss_cornernet(torch.from_numpy(g["pred_hm"]), torch.from_numpy(g["gt_hm"]))
    np.testing.assert_allclose(focal.numpy(), g["focal"], rtol=1e-5)
    out = torch.from_numpy(g["reg_out"])
    b, d, h, w = out.shape
    rows = out.permute(0, 2, 3, 1).reshape(b, h * w, d)
    pred = torch.gather(rows, 1, torch.from_numpy(g["reg_inds"])[..., None].expand(-1, -1, d))
    rl = cl.reg_loss(pred, torch.from_numpy(g["reg_tgt"]), torch.from_numpy(g["reg_mask"]))
    np.testing.assert_allclose(rl.numpy(), g["reg_loss"], rtol=1e-5, atol=1e-6)


def test_center_head_loss_rows_equal_nchw_form(golden):
    """The channels-last row form used by the train engine == the NCHW composition of get_loss."""
    g = golden("center_loss")
    torch.manual_seed(0)
    B, H, W = 2, 24, 20
    rows = torch.randn(B * H * W, 16, requires_grad=True)
    gt_hm = torch.from_numpy(g["gt_hm"])
    inds, mask, tgt = torch.from_numpy(g["reg_inds"]), torch.from_numpy(g["reg_mask"]), torch.from_numpy(g["reg_tgt"])
    loss, parts = cl.center_head_loss(rows, B, H, W, gt_hm, tgt, inds, mask)
    maps = rows.view(B, H, W, 16).permute(0, 3, 1, 2)
    hm = torch.clamp(maps[:, 8:11].sigmoid(), 1e-4, 1 - 1e-4)
    want_hm = cl.neg_loss_cornernet(hm, gt_hm)
    r = maps[:, 0:8].permute(0, 2, 3, 1).reshape(B, H * W, 8)
    pred = torch.gather(r, 1, inds[..., None].expand(-1, -1, 8))
    want = want_hm + cl.reg_loss(pred, tgt, mask).sum() * 2.0
    np.testing.assert_allclose(loss.item(), want.item(), rtol=1e-6)
    loss.backward()
    assert rows.grad[:, 11:].abs().max() == 0 and rows.grad[:, :11].abs().sum() > 0


def test_assign_targets_filters_before_truncating(golden):
    """center_head.py:180-196 keeps a head's boxes first and l.113 then walks the first NUM_MAX_OBJS of THAT list: with
    padding rows interleaved and more boxes than slots, the result is that of the compacted list."""
    g = golden("center_loss")
    real = torch.from_numpy(g["gt_boxes"])[:12]
    assert (real[:, 7] >= 1).all()
    mixed = torch.zeros((24, 8))
    mixed[1::2] = real                                       # padding row, box, padding row, box, ...
    args = ((188, 188), [-75.2, -75.2, -2, 75.2, 75.2, 4], [0.1, 0.1, 0.15], 3)
    kw = dict(feature_map_stride=8, num_max_objs=10, gaussian_overlap=0.1, min_radius=2)
    got = cl.assign_targets(mixed[None], *args, **kw)
    want = cl.assign_targets(real[None], *args, **kw)       # already compact: its first 10 boxes
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert int(got[3].sum()) == 10
