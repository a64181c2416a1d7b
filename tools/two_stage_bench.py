#!/usr/bin/env python3
"""Timing of the two-stage engine alone (what bench.py reports as value_two_stage).  FRAMES=16 python tools/two_stage_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import models
from cpd_amd.engine import ModelConfig, init_state_dict
from cpd_amd.two_stage import VoxelRCNNEngine
from cpd_amd.synthetic import waymo_cloud
B = int(os.environ.get("FRAMES", "16"))
cfg = ModelConfig(); sd = init_state_dict(cfg, 0)
mcfg = models.waymo_voxel_rcnn_cfg()
torch.manual_seed(0)
head = models.__all__[mcfg.ROI_HEAD.NAME](input_channels={"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 128}, model_cfg=mcfg.ROI_HEAD,
                                          point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size, num_class=1)
sd.update({"roi_head." + k: v.detach().clone() for k, v in head.state_dict().items()})
eng = VoxelRCNNEngine(cfg, mcfg.ROI_HEAD, mcfg.POST_PROCESSING, sd, host_results=True)
clouds = [torch.from_numpy(waymo_cloud(s)).cuda() for s in range(B)]
for _ in range(2):
    res, it = eng.forward(clouds, return_intermediates=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4):
    eng.forward(clouds)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
t1 = time.perf_counter()
for _ in range(4):
    eng.rpn.forward(clouds, proposals=eng.sources)
torch.cuda.synchronize(); d1 = (time.perf_counter() - t1) / 4
print("two-stage %.1f frames/s (%.2f ms/step of %d frames); first stage alone %.2f ms; rois/frame %d; final boxes/frame %.1f" % (
    B / dt, 1e3 * dt, B, 1e3 * d1, it["rois"].shape[1], sum(len(r["pred_boxes"]) for r in res) / B))
