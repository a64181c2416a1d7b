"""First stage of the OTHER shipped two-stage family: `voxel_rcnn_dbscan_single_train.yaml` / `voxel_rcnn_oyster_single_train.yaml`
(tools/cfgs/models/waymo_unsupervised/, l.12 `NAME: VoxelRCNN`, l.37 `DENSE_HEAD.NAME: AnchorHeadSingleV2`, l.95 `ROI_HEAD.NAME:
VoxelRCNNHead`) as a fused engine: the CenterPoint engine's voxelizer / sparse backbone / BEV backbone (cpd_amd/engine.py) with
`AnchorHeadSingleV2` (anchor_head_single.py:31-192) as the dense head and the RoI head's `proposal_layer`
(roi_head_template.py:53-114, NMS_CONFIG.TEST) as the stage's last step:

    points -> voxelize -> VoxelResBackBone8x -> HeightCompression -> BaseBEVBackbone            (CenterPointEngine, unchanged)
           -> shared 3x3 conv, five get_layer branches, direction classifier                      (AnchorHeadSingleV2._heads_eval: four launches)
           -> occupancy anchor mask, generate_predicted_boxes (cpd_anchor_decode)                 (anchor_head.py)
           -> per-sample max class score, top NMS_PRE_MAXSIZE, rotated NMS, first NMS_POST_MAXSIZE (roi_pool.proposal_layer: batched launches)

`forward(points_list, proposals=levels)` returns what `CenterPointEngine.forward(proposals=...)` returns -- the padded RoI block, scores,
1-based labels, per-frame counts (the step's one read-back after the voxel / level counts) and the named levels -- so
`two_stage.VoxelRCNNEngine(..., rpn=AnchorPointEngine(...))` is the fused detector of those configs. The dense half keeps fp32 maps here
(the head's 1 x 1 GEMMs have 60 / 12 output columns: the fp32 wave kernels read fp32 rows)."""
from typing import Dict

import numpy as np
import torch

from . import anchor_head, ops, roi_pool
from .engine import CenterPointEngine, ModelConfig


class AnchorPointEngine(CenterPointEngine):
    pad_label = 1                          # roi_head_template.py:111: `roi_labels + 1` on every slot of the zero buffer, padded ones included

    def __init__(self, cfg: ModelConfig, state_dict: Dict[str, torch.Tensor], head_cfg, nms_cfg, class_names=("Vehicle", "Pedestrian", "Cyclist"),
                 device="cuda", host_results=False):
        self.head_cfg, self.nms_cfg, self.class_names = head_cfg, nms_cfg, list(class_names)
        self.proposal_first_rows = None       # candidates per frame the proposal NMS looks at first (None: max(512, 4 * NMS_POST_MAXSIZE))
        super().__init__(cfg, state_dict, device=device, host_results=host_results)
        if nms_cfg.get("MULTI_CLASSES_NMS", False) or nms_cfg["NMS_TYPE"] != "nms_gpu":
            raise NotImplementedError("proposal NMS variant not selected by the shipped CPD configs")

    # ------------------------------------------------------------------ the dense head
    def _build_head(self):
        cfg = self.cfg
        grid = np.array(ops.voxel_grid_size(cfg.voxel_size, cfg.point_cloud_range)[::-1])            # x, y, z like dataset.grid_size
        head = anchor_head.AnchorHeadSingleV2(self.head_cfg, input_channels=sum(cfg.bev_num_upsample_filters), num_class=cfg.num_class,
                                              class_names=self.class_names, grid_size=grid, point_cloud_range=cfg.point_cloud_range,
                                              conv_math=cfg.conv_math)
        own = head.state_dict()
        sub = {k[len("dense_head."):]: v for k, v in self.sd.items() if k.startswith("dense_head.")}
        missing = [k for k in own if k not in sub]
        if missing:
            raise KeyError("AnchorPointEngine: dense_head parameters missing from the state dict: %s ..." % missing[:4])
        head.load_state_dict({k: sub[k] for k in own})
        self.head = head.to(self.device).eval()

    def _head_pairs_ok(self, batch, h, w, c_cat):
        return False                                   # fp32 dense maps (module docstring)

    def _head_rows(self, cat, batch, h, w, T, pairs):
        from .models import _nchw
        assert not pairs
        return self.head._heads_eval(_nchw(cat, batch, h, w))                  # (cls rows, box rows, dir rows | None): four launches

    # ------------------------------------------------------------------ anchors -> proposals
    def decode_and_nms(self, head_rows, batch, h, w, raw=False):
        """AnchorHeadSingleV2.forward's tail (anchor_head_single.py:160-190: occupancy mask, masked anchors, generate_predicted_boxes)
        + RoIHeadTemplate.proposal_layer (roi_head_template.py:53-114). Only as a first stage (`raw`): the one-stage anchor
        detector is not one of the shipped configs."""
        if not raw:
            raise NotImplementedError("AnchorPointEngine is a first stage: call forward(points, proposals=[...])")
        hd = self.head
        cls_r, box_r, dir_r = head_rows
        if hd.anchors_root is None:
            hd.anchors_root = hd._gen[0].generate_anchors(hd._gen[1], device=self.device)[0]
        xy = torch.cat([p[:, :2] for p in self._cur_points])
        pts = torch.nn.functional.pad(xy, (1, 0))                                # get_anchor_mask reads columns 1, 2 of (b, x, y, ...) points
        mask = hd.get_anchor_mask(pts, (h, w))
        anchors = [a[:, mask, ...] for a in hd.anchors_root]
        pick = lambda r: r.reshape(batch, h, w, r.shape[1])[:, mask, :]
        cls, boxes = anchor_head.generate_predicted_boxes(anchors, batch, pick(cls_r), pick(box_r), pick(dir_r) if dir_r is not None else None,
                                                          self.head_cfg.get("DIR_OFFSET", 0.78539), self.head_cfg.get("DIR_LIMIT_OFFSET", 0.0),
                                                          hd.num_dir_bins)
        self.last_dense = dict(batch_cls_preds=cls, batch_box_preds=boxes, anchor_mask=mask)      # (references, for tests and tools)
        nms = self.nms_cfg
        post = int(nms["NMS_POST_MAXSIZE"])
        # proposal NMS over each frame's first `rows` candidates (exact while the NMS_POST_MAXSIZE-th survivor is among them: it almost
        # always is -- a 0.8 threshold suppresses little -- and the `incomplete` word that says otherwise rides with the counts)
        rows = min(int(nms["NMS_PRE_MAXSIZE"]), max(512, 64 * ((4 * post + 63) // 64))) if self.proposal_first_rows is None else int(self.proposal_first_rows)
        rois, scores, labels, kept, inc = roi_pool.proposal_layer(boxes, cls, float(nms["NMS_THRESH"]), int(nms["NMS_PRE_MAXSIZE"]), post,
                                                                  first_rows=rows)
        flag = self._range_exceeded_flag()
        tail = [inc.max().view(1)] + ([flag] if flag is not None else [])
        ns = torch.cat([kept.to(torch.int32)] + tail).tolist()                                       # the stage's one read-back
        if ns[batch]:                                    # some frame needs boxes beyond its first `rows`: the full NMS, once more
            self.proposal_full_reruns = getattr(self, "proposal_full_reruns", 0) + 1
            rois, scores, labels, kept = roi_pool.proposal_layer(boxes, cls, float(nms["NMS_THRESH"]), int(nms["NMS_PRE_MAXSIZE"]), post)
            ns[:batch] = kept.tolist()
        del ns[batch]
        high = bool(ns[batch]) if flag is not None else False
        if getattr(self, "_rb_scaled", False):
            self._range_exceeded, self._range_high = False, high
        else:
            self._range_exceeded = high
        if self._range_exceeded:
            return None
        return rois, scores, labels, [int(v) for v in ns[:batch]]

    @torch.no_grad()
    def forward(self, points_list, return_intermediates=False, proposals=None, pair_levels=False):
        if isinstance(points_list, torch.Tensor):
            points_list = [points_list]
        if proposals is None:
            raise NotImplementedError("AnchorPointEngine is a first stage: call forward(points, proposals=[...])")
        self._cur_points = points_list
        try:
            return super().forward(points_list, return_intermediates=False, proposals=proposals, pair_levels=pair_levels)
        finally:
            self._cur_points = None

    __call__ = forward


def synthetic_two_stage_state(cfg: ModelConfig, sd, seed=1):
    """Random-init parameters of the dbscan / oyster model on top of a first-stage state dict `sd` (engine.init_state_dict): the
    AnchorHeadSingleV2 dense head -- its branch convs and class scores SPREAD (the reference's N(0, 0.001) init leaves every anchor at its
    bias: nothing to rank) -- and a class-agnostic VoxelRCNNHead. What bench.py's `value_two_stage_anchor` and the full-size parity test
    (tests/test_gpu_two_stage.py) both run. -> (model cfg, state dict)"""
    from . import models
    mcfg = models.waymo_voxel_rcnn_dbscan_cfg()
    torch.manual_seed(seed)
    grid = np.array(ops.voxel_grid_size(cfg.voxel_size, cfg.point_cloud_range)[::-1])
    dh = anchor_head.AnchorHeadSingleV2(dbscan_dense_head_cfg(), input_channels=sum(cfg.bev_num_upsample_filters), num_class=cfg.num_class,
                                        class_names=["Vehicle", "Pedestrian", "Cyclist"], grid_size=grid, point_cloud_range=cfg.point_cloud_range)
    with torch.no_grad():
        for br in dh.BRANCHES:
            getattr(dh, br)[0].weight.normal_(0, (2.0 / (9 * 64)) ** 0.5)
        dh.conv_cls[3].weight.normal_(0, 0.5)
    out = {k: v for k, v in sd.items() if not k.startswith("dense_head.")}
    out.update({"dense_head." + k: v.detach().clone() for k, v in dh.state_dict().items()})
    out.update({"roi_head." + k: v.detach().clone() for k, v in
                models.__all__["VoxelRCNNHead"](input_channels={"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 128},
                                                 model_cfg=mcfg.ROI_HEAD, point_cloud_range=cfg.point_cloud_range,
                                                 voxel_size=cfg.voxel_size, num_class=1).state_dict().items()})
    return mcfg, out


def dbscan_dense_head_cfg():
    """DENSE_HEAD of voxel_rcnn_dbscan_single_train.yaml:36-93 (the oyster config's is the same)."""
    def cls(name, size, un):
        return dict(class_name=name, anchor_sizes=[size], anchor_rotations=[0, 1.57], anchor_bottom_heights=[0], align_center=False,
                    feature_map_stride=8, matched_threshold=0.55, unmatched_threshold=un)
    return dict(NAME="AnchorHeadSingleV2", CLASS_AGNOSTIC=False, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539, DIR_LIMIT_OFFSET=0.0,
                NUM_DIR_BINS=2,
                ANCHOR_GENERATOR_CONFIG=[cls("Vehicle", [4.7, 2.1, 1.7], 0.5), cls("Pedestrian", [0.91, 0.86, 1.73], 0.4),
                                         cls("Cyclist", [1.78, 0.84, 1.78], 0.4)],
                TARGET_ASSIGNER_CONFIG=dict(NAME="AxisAlignedTargetAssigner", POS_FRACTION=-1.0, SAMPLE_SIZE=512, NORM_BY_NUM_EXAMPLES=False,
                                            MATCH_HEIGHT=False, BOX_CODER="ResidualCoder"),
                LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=[1.0] * 7)))
