#!/usr/bin/env python3
"""Timing of the anchor-head two-stage engine alone (bench.py's value_two_stage_anchor).  FRAMES=16 python tools/anchor_two_stage_bench.py
(under rocprofv3 --kernel-trace --stats for the kernel table: tools/kstats-like use)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cpd_amd import anchor_head, models, ops
from cpd_amd.anchor_engine import AnchorPointEngine, dbscan_dense_head_cfg
from cpd_amd.engine import ModelConfig, init_state_dict
from cpd_amd.two_stage import VoxelRCNNEngine
from cpd_amd.synthetic import waymo_cloud
B = int(os.environ.get("FRAMES", "16"))
cfg = ModelConfig(); sd = init_state_dict(cfg, 0)
mcfg = models.waymo_voxel_rcnn_dbscan_cfg()
torch.manual_seed(1)
grid = np.array(ops.voxel_grid_size(cfg.voxel_size, cfg.point_cloud_range)[::-1])
dh = anchor_head.AnchorHeadSingleV2(dbscan_dense_head_cfg(), input_channels=sum(cfg.bev_num_upsample_filters), num_class=cfg.num_class,
                                    class_names=["Vehicle", "Pedestrian", "Cyclist"], grid_size=grid, point_cloud_range=cfg.point_cloud_range)
with torch.no_grad():
    for br in dh.BRANCHES:
        getattr(dh, br)[0].weight.normal_(0, (2.0 / (9 * 64)) ** 0.5)
    dh.conv_cls[3].weight.normal_(0, 0.5)
a_sd = {k: v for k, v in sd.items() if not k.startswith("dense_head.")}
a_sd.update({"dense_head." + k: v.detach().clone() for k, v in dh.state_dict().items()})
a_sd.update({"roi_head." + k: v.detach().clone() for k, v in
             models.__all__["VoxelRCNNHead"](input_channels={"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 128}, model_cfg=mcfg.ROI_HEAD,
                                              point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size, num_class=1).state_dict().items()})
rpn = AnchorPointEngine(cfg, a_sd, mcfg.DENSE_HEAD, mcfg.ROI_HEAD.NMS_CONFIG["TEST"])
two = VoxelRCNNEngine(cfg, mcfg.ROI_HEAD, mcfg.POST_PROCESSING, a_sd, host_results=True, rpn=rpn)
clouds = [torch.from_numpy(waymo_cloud(s)).cuda() for s in range(B)]
for _ in range(2):
    two.forward(clouds)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4):
    two.forward(clouds)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
t1 = time.perf_counter()
for _ in range(4):
    rpn.forward(clouds, proposals=two.sources)
torch.cuda.synchronize(); d1 = (time.perf_counter() - t1) / 4
print("anchor two-stage %.1f frames/s (%.2f ms/step of %d frames); first stage alone %.2f ms" % (B / dt, 1e3 * dt, B, 1e3 * d1))
