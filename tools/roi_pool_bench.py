#!/usr/bin/env python3
"""RoI grid pooling on the hot path's own multi-scale features: time per stage for
B frames x N rois (216 grid points each), dense voxel2pinds volume vs site-index query."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpd_amd import ops, roi_pool
from cpd_amd.engine import CenterPointEngine, ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = ModelConfig()
eng = CenterPointEngine(cfg, init_state_dict(cfg, seed=0))
pts = [torch.from_numpy(waymo_cloud(i)).cuda() for i in range(B)]
res, it = eng.forward(pts, return_intermediates=True)
rois = torch.zeros((B, N, 7), device="cuda")
for b in range(B):
    k = min(N, res[b]["pred_boxes"].shape[0])
    rois[b, :k] = res[b]["pred_boxes"][:k]
    rois[b, k:, 3:6] = 1.0
torch.manual_seed(0)
layers = {name: roi_pool.NeighborVoxelSAModuleMSG(query_ranges=[[2, 2, 2], [4, 4, 4]], radii=r, nsamples=[16, 16],
                                                  mlps=[[c, 32, 32], [c, 32, 32]]).cuda().eval()
          for name, r, c in (("x_conv3", [0.4, 0.8], 64), ("x_conv4", [0.8, 1.6], 128))}
strides = {"x_conv3": 4, "x_conv4": 8}
indexes = {}
for name in layers:
    f, c, s = it["levels"][name]
    indexes[name] = ops.SiteIndex.build(c, B, s)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


args = (rois, it["levels"], strides, layers, 6, cfg.voxel_size, cfg.point_cloud_range, B)
t_dense = timed(lambda: roi_pool.roi_grid_pool(*args))
t_index = timed(lambda: roi_pool.roi_grid_pool(*args, indexes=indexes))
m = B * N * 216
# the query alone, widest window (9^3 cells), level x_conv3
f, c, s = it["levels"]["x_conv3"]
xyz = roi_pool.get_voxel_centers(c[:, 1:4], 4, cfg.voxel_size, cfg.point_cloud_range).contiguous()
grid, _ = roi_pool.get_global_grid_points_of_roi(rois, 6)
gx = grid.view(-1, 3).contiguous()
gc = torch.cat([torch.arange(B, device="cuda").repeat_interleave(N * 216)[:, None].float(),
                ((gx[:, 2:3] - cfg.point_cloud_range[2]) // cfg.voxel_size[2]) // 4,
                ((gx[:, 1:2] - cfg.point_cloud_range[1]) // cfg.voxel_size[1]) // 4,
                ((gx[:, 0:1] - cfg.point_cloud_range[0]) // cfg.voxel_size[0]) // 4], 1).int().contiguous()
v2p = roi_pool.generate_voxel2pinds(c, B, s)
tq_dense = timed(lambda: roi_pool.voxel_query([4, 4, 4], 0.8, 16, xyz, gx, gc, point_indices=v2p))
tq_index = timed(lambda: roi_pool.voxel_query([4, 4, 4], 0.8, 16, xyz, gx, gc, index=indexes["x_conv3"]))
tv2p = timed(lambda: roi_pool.generate_voxel2pinds(c, B, s))
print("B %d x %d rois: %d grid points; roi_grid_pool (2 levels x 2 scales) dense-volume %.2f ms, site-index %.2f ms" % (B, N, m, t_dense, t_index))
print("x_conv3 (%d voxels, grid %s): voxel2pinds %.1f us (%.0f MB volume); 9^3-cell query dense %.1f us, index %.1f us "
      "(%.1f G cell tests/s)" % (c.shape[0], s, tv2p * 1e3, v2p.numel() * 4 / 1e6, tq_dense * 1e3, tq_index * 1e3,
                                 m * 729 / (tq_index * 1e-3) / 1e9))

# whole second stage (reference cfg sizes: GRID_SIZE 6, MLPS [[32,32],[32,32]] per level, SHARED_FC/CLS_FC/REG_FC [256,256])
head_cfg = dict(ROI_GRID_POOL=dict(FEATURES_SOURCE=["x_conv3", "x_conv4"], GRID_SIZE=6, POOL_LAYERS=dict(
    x_conv3=dict(MLPS=[[32, 32], [32, 32]], QUERY_RANGES=[[2, 2, 2], [4, 4, 4]], POOL_RADIUS=[0.4, 0.8], NSAMPLE=[16, 16], POOL_METHOD="max_pool"),
    x_conv4=dict(MLPS=[[32, 32], [32, 32]], QUERY_RANGES=[[2, 2, 2], [4, 4, 4]], POOL_RADIUS=[0.8, 1.6], NSAMPLE=[16, 16], POOL_METHOD="max_pool"))),
    SHARED_FC=[256, 256], CLS_FC=[256, 256], REG_FC=[256, 256], DP_RATIO=0.3)
head = roi_pool.VoxelRCNNHead({"x_conv3": 64, "x_conv4": 128}, head_cfg, point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size).cuda().eval()
bd = {"batch_size": B, "rois": rois, "multi_scale_3d_features": it["levels"], "multi_scale_3d_strides": strides}
t_head = timed(lambda: head(dict(bd)))
print("VoxelRCNNHead eval forward (pool + 27648->256->256 shared FC + cls/reg stacks + decode), B %d x %d rois: %.2f ms" % (B, N, t_head))
