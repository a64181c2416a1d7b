"""The oracle's SECOND-stage restatement (tests/ref_two_stage.py: numpy + oracle.voxel_query / anchor_decode / nms) pinned on the
reference's own VoxelRCNNHead output (tests/golden/voxel_rcnn_head.npz, produced by tests/golden/make_golden.py from the reference
module) -- so that test_gpu_two_stage.py's full-size engine-vs-oracle comparison rests on a checker that itself reproduces the
reference. CPU only."""
import numpy as np

import ref_two_stage as r2


def test_second_stage_restatement_reproduces_the_reference_module(oracle, golden):
    g = golden("voxel_rcnn_head")
    roi_cfg = dict(ROI_GRID_POOL=dict(FEATURES_SOURCE=["x_conv3", "x_conv4"], GRID_SIZE=3, POOL_LAYERS=dict(
        x_conv3=dict(MLPS=[[16, 16], [16, 16]], QUERY_RANGES=[[1, 1, 1], [2, 2, 2]], POOL_RADIUS=[0.6, 1.2], NSAMPLE=[8, 8], POOL_METHOD="max_pool"),
        x_conv4=dict(MLPS=[[16, 16], [16, 16]], QUERY_RANGES=[[1, 1, 1], [2, 2, 2]], POOL_RADIUS=[1.2, 2.4], NSAMPLE=[8, 8], POOL_METHOD="max_pool"))),
        SHARED_FC=[64, 64], CLS_FC=[32], REG_FC=[32], DP_RATIO=0.3)
    sd = {"roi_head." + k[3:]: np.asarray(v) for k, v in g.items() if k.startswith("rh.")}
    levels = {"x_conv3": (g["c3_feat"], g["c3_idx"], [11, 104, 104]), "x_conv4": (g["c4_feat"], g["c4_idx"], [5, 52, 52])}
    rois = g["rois"]

    class Cfg:
        voxel_size = [0.1, 0.1, 0.15]
        point_cloud_range = g["pcr"].tolist()
    post = dict(SCORE_THRESH=0.01, NMS_CONFIG=dict(NMS_THRESH=0.3, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500))
    labels = np.ones(rois.shape[:2], np.int64)
    final, it = r2.second_stage(oracle, Cfg, roi_cfg, post, sd, rois, labels, levels, rois.shape[0])
    np.testing.assert_allclose(it["batch_cls_preds"], g["batch_cls_preds"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(it["batch_box_preds"], g["batch_box_preds"], atol=2e-4, rtol=1e-5)
    assert len(final) == rois.shape[0] and all(len(f["pred_boxes"]) > 0 for f in final)
