"""Training through the drop-in modules with the levels in tap-pattern row order (spconv.install(row_order="taps")) against canonical
order -- and against a CONTROL: the same voxels handed over in another row order, levels canonical. Training-mode BatchNorm sums batch
statistics over the rows in their order, so any re-ordering moves activations by fp32 rounding (1e-6 relative), a few of 10^7 ReLUs sit
within that of the kink, and each flip moves parameter gradients by percents of their maximum: the tap order must stay within what the
control shows. Full-size cloud, two frames, f32.   python tools/taps_train_check.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import models, ops
from cpd_amd import spconv as sp
from cpd_amd.engine import ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud, gt_boxes
cfg = ModelConfig(); sd = init_state_dict(cfg, seed=4)
vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, cfg.num_point_features, cfg.max_points_per_voxel, cfg.max_voxels)
clouds = [torch.from_numpy(waymo_cloud(s)).cuda() for s in (0, 1)]
gt = torch.stack([torch.from_numpy(gt_boxes(s)) for s in (0, 1)]).cuda()


def run(order, shuffle=False):
    sp.install(conv_math="f32", row_order=order)
    net = models.CenterPoint(point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size).cuda().train()
    net.load_state_dict(sd)
    _, coords, _, feats, nvox = vox.batch(clouds)
    n = int(nvox[2])
    f_in, c_in = feats[:n].clone(), coords[:n].clone()
    if shuffle:                                                   # rows of frame b stay frame b's
        perm = torch.argsort(c_in[:, 0].long() * (1 << 40) + torch.randperm(n, device=c_in.device))
        f_in, c_in = f_in[perm].contiguous(), c_in[perm].contiguous()
    bd = {"voxel_features": f_in, "voxel_coords": c_in, "batch_size": 2, "gt_boxes": gt}
    ret, _, _ = net(bd)
    ret["loss"].backward()
    return float(ret["loss"].detach()), bd["spatial_features"].detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


def worst(ga, gb):
    # (a conv bias in front of a BatchNorm has a mathematically zero gradient -- rounding noise, ~1e-6 against O(1) elsewhere: differences
    # are measured against the larger of the tensor's own scale and 1e-3 of the largest gradient)
    gmax = max(float(v.abs().max()) for v in ga.values())
    rel = {k: float((ga[k] - gb[k]).abs().max() / max(float(ga[k].abs().max()), 1e-3 * gmax)) for k in ga}
    k = max(rel, key=rel.get)
    return rel[k], k


torch.manual_seed(0)
base, again, taps, ctrl = run("canonical"), run("canonical"), run("taps"), run("canonical", shuffle=True)
sp.install(row_order="canonical")
assert again[0] == base[0] and all(torch.equal(base[2][k], again[2][k]) for k in base[2]), "the same run twice is not bitwise repeatable"
for name, r in (("tap-pattern levels", taps), ("control: shuffled input rows", ctrl)):
    w, k = worst(base[2], r[2])
    print("%-30s loss %.6f (canonical %.6f)  BEV map max |diff| %.2e of %.1f  worst gradient difference %.2e of its maximum (%s)" % (
        name, r[0], base[0], float((r[1] - base[1]).abs().max()), float(base[1].abs().max()), w, k))
wt, wc = worst(base[2], taps[2])[0], worst(base[2], ctrl[2])[0]
assert abs(taps[0] - base[0]) <= 1e-5 * abs(base[0]) and wt <= 3.0 * wc + 1e-3, (wt, wc)
print("taps training ok: within the control's spread")
