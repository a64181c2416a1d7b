"""Fused inference engine of the TWO-STAGE detector the shipped CPD config selects (`NAME: VoxelRCNN`,
tools/cfgs/models/waymo_unsupervised/voxel_rcnn_cproto_center.yaml:13): one process per GPU, a flat list of C-ABI launches.

    VoxelRCNN.forward (cpd/models/detectors/voxel_rcnn.py:8-27), eval:
      vfe -> backbone_3d -> map_to_bev -> backbone_2d -> dense_head          = CenterPointEngine (cpd_amd/engine.py), whose
                                                                               NMS output is the second stage's `rois`
                                                                               (center_head.py:340-350, reorder_rois_for_refining)
      roi_head (VoxelRCNNProtoHead eval branch, voxel_rcnn_head.py:581-637):
          roi_grid_pool on x_conv3 / x_conv4 (l.186-273)                       = cpd_voxel_query_index + cpd_voxel_pool_max + 1x1
                                                                               cpd_gather_conv GEMMs (cpd_amd/roi_pool.py)
          shared_fc / cls / reg stacks (Linear + BatchNorm1d + ReLU)           = cpd_gather_conv GEMMs, BatchNorm folded
          generate_predicted_boxes (roi_head_template.py:269-299)              = cpd_anchor_decode + rotation
      post_processing (detector3d_template.py:222-343): sigmoid, RoI labels,   = batched on the device: one sort, cpd_nms_batch,
          class-agnostic NMS (model_nms_utils.py:113-134)                        cpd_select_boxes, ONE count read-back

Host synchronisations per step: the voxel count, the per-level row counts, the first stage's per-frame box counts (they size the
RoI block exactly as the reference does: padded with zero boxes to the batch maximum) and the final counts.
Weights come in under the reference's state_dict names (`roi_head.*` next to the first stage's); the prototype branch's parameters
(`*_mm`, `*_P`: training only) are ignored."""
from typing import Dict, List

import torch

from . import ops, roi_pool
from .engine import CenterPointEngine, ModelConfig


class VoxelRCNNEngine:
    """points [N, C] per frame -> {'pred_boxes', 'pred_scores', 'pred_labels'} per frame (second-stage refined, final NMS 0.3)."""

    def __init__(self, cfg: ModelConfig, roi_cfg, post_cfg, state_dict: Dict[str, torch.Tensor], device="cuda", host_results=False, rpn=None):
        """`rpn`: the first stage -- any engine whose forward(points_list, proposals=levels) returns (RoI block, scores, 1-based labels,
        per-frame counts, levels) -- with `pair_levels=True` the levels may come back as cpd_amd.ops.PairRows-tagged fp16-pair rows -- and that
        keeps `level_indexes` and `pad_label` (the label of a padded RoI slot: 0 for CenterHead, 1 for proposal_layer); default the CenterPoint engine on the same state dict
        (voxel_rcnn_cproto_center.yaml); cpd_amd.anchor_engine.AnchorPointEngine for the dbscan / oyster configs."""
        self.cfg, self.roi_cfg, self.post_cfg = cfg, roi_cfg, post_cfg
        self.device = torch.device(device)
        self.host_results = bool(host_results)
        self.rpn = rpn if rpn is not None else CenterPointEngine(cfg, state_dict, device=device)
        nf = cfg.num_filters
        channels = {"x_conv1": nf[0], "x_conv2": nf[1], "x_conv3": nf[2], "x_conv4": nf[3]}
        agnostic = bool(roi_cfg.get("CLASS_AGNOSTIC", False))
        head = roi_pool.VoxelRCNNHead(channels, roi_cfg, point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size,
                                      num_class=1 if agnostic else cfg.num_class)
        own = head.state_dict()
        sub = {k[len("roi_head."):]: v for k, v in state_dict.items() if k.startswith("roi_head.")}
        missing = [k for k in own if k not in sub]
        if missing:
            raise KeyError("VoxelRCNNEngine: roi_head parameters missing from the state dict: %s ..." % missing[:4])
        head.load_state_dict({k: sub[k] for k in own})
        self.head = head.to(self.device).eval()
        self.head._pack_fc()
        for layer in self.head.roi_grid_pool_layers:
            layer._pack()
        self.sources = list(self.head.sources)
        self.strides = {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}
        nms = post_cfg["NMS_CONFIG"]
        if nms["MULTI_CLASSES_NMS"] or post_cfg.get("WBF", False) or nms["NMS_TYPE"] != "nms_gpu":
            raise NotImplementedError("post-processing variant not selected by the shipped CPD configs")

    @torch.no_grad()
    def forward(self, points_list: List[torch.Tensor], return_intermediates=False):
        if isinstance(points_list, torch.Tensor):
            points_list = [points_list]
        batch = len(points_list)
        # ---- first stage: proposals (padded device block + host counts) and the pooled levels
        # (the pooled levels stay fp16-pair rows when the first stage ran on them: the pooling's first 1 x 1 GEMM takes the stored bits;
        # intermediates for a caller are fp32)
        ob, os_, ol, counts, levels = self.rpn.forward(points_list, proposals=self.sources, pair_levels=not return_intermediates)
        n_roi = max(1, max(counts))                                     # reorder_rois_for_refining: at least one (zero) RoI
        cnt = torch.tensor(counts, dtype=torch.int64, device=self.device)
        valid = torch.arange(n_roi, device=self.device)[None, :] < cnt[:, None]
        # zero boxes past a frame's count, like the reference's new_zeros block -- by SELECTION: the slots past a frame's count were
        # never written by cpd_select_boxes (stale allocator bytes, possibly NaN bit patterns, and NaN * 0 is NaN: ADVICE r4)
        rois = torch.where(valid[..., None], ob[:, :n_roi], ob.new_zeros(())).contiguous()
        # the label of a padded slot is the first stage's: 0 behind CenterHead's reorder_rois_for_refining (center_head.py:340-350, a
        # new_zeros block), 1 behind RoIHeadTemplate.proposal_layer (roi_head_template.py:111 adds 1 to EVERY slot of its zero buffer)
        roi_labels = torch.where(valid, ol[:, :n_roi], ol.new_full((), int(getattr(self.rpn, "pad_label", 0)))).contiguous()
        # ---- second stage
        lv = {name: levels[name] for name in self.sources}
        m = self.cfg.conv_math if self.cfg.conv_math == "f16x2" else None     # FC stacks on the split-fp16 tile kernels, range-guarded
        xb = ops.absmax_blocks(1, self.device)[0] if m else None              # the pooled rows' range block, raised by the pooling kernels themselves
        pooled = roi_pool.roi_grid_pool(rois, lv, self.strides, dict(zip(self.sources, self.head.roi_grid_pool_layers)), self.head.grid_size,
                                        self.cfg.voxel_size, self.cfg.point_cloud_range, batch,
                                        indexes={name: self.rpn.level_indexes[name] for name in self.sources}, out_block=xb)
        x = pooled.reshape(pooled.shape[0], -1).contiguous()
        fc = self.head._fc
        shared, rb = self.head._run(fc["shared_fc_layers"], x, math=m, in_block=xb, return_block=True)    # rb: the block the last shared layer's epilogue filled
        rcnn_cls = self.head._run(fc["cls_layers"], shared, math=m, in_block=rb)
        rcnn_reg = self.head._run(fc["reg_layers"], shared, math=m, in_block=rb)
        cls, boxes = self.head.generate_predicted_boxes(batch, rois, rcnn_cls, rcnn_reg)
        # ---- post_processing, all frames at once
        pp, nms = self.post_cfg, self.post_cfg["NMS_CONFIG"]
        # sigmoid, max over classes, threshold, stable descending rank, gathers: one launch (cpd_rank_scores)
        sboxes, ranked, slabels, n_ok = ops.rank_scores(cls, boxes, roi_labels, pp["SCORE_THRESH"], int(nms["NMS_PRE_MAXSIZE"]))
        keep, num_keep = ops.nms_batch(sboxes, n_ok, float(nms["NMS_THRESH"]))
        post = min(int(nms["NMS_POST_MAXSIZE"]), n_roi)
        fb, fs, fl, fn, blk = ops.select_boxes(sboxes, ranked, slabels, keep, num_keep, post, label_offset=0, packed=True)
        if self.host_results:                                            # counts + boxes + scores + labels: one block, one copy
            hdr, fb, fs, fl = ops.unpack_boxes(blk.cpu(), blk._cpd_layout)
            ns = hdr.tolist()
        else:
            ns = fn.tolist()                                             # the stage's one read-back
        out = [{"pred_boxes": fb[b, :ns[b]], "pred_scores": fs[b, :ns[b]], "pred_labels": fl[b, :ns[b]]} for b in range(batch)]
        if return_intermediates:
            return out, dict(rois=rois, roi_labels=roi_labels, roi_scores=torch.where(valid, os_[:, :n_roi], os_.new_zeros(())), batch_box_preds=boxes,
                             batch_cls_preds=cls, levels=lv, pooled=pooled)
        return out

    __call__ = forward
