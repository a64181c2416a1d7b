"""Fused forward engine of the hot path: points -> boxes, one process per GPU.

Mirrors CenterPoint.forward's module_topology (cpd/models/detectors/detector3d_template.py:22-25,
centerpoint.py:9-22) for the one-stage path
    MeanVFE -> VoxelResBackBone8x -> HeightCompression -> BaseBEVBackbone -> CenterHead -> NMS
but as a flat list of C-ABI launches on one HIP stream: eval-mode BatchNorm, bias, ReLU and the
SparseBasicBlock residual are folded into the conv epilogues, activations stay channels-last in
HBM, rulebooks are built once per indice_key, and nothing leaves the device until the final boxes.
Weights come in under the reference's own state_dict names (so a CPD checkpoint loads unchanged).
"""
import contextlib
import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch

from . import _lib, ops


@dataclass
class ModelConfig:
    """The fields of voxel_rcnn_cproto_center.yaml / waymo_unsupervised_cproto.yaml the path reads."""
    point_cloud_range: List[float] = field(default_factory=lambda: [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0])
    voxel_size: List[float] = field(default_factory=lambda: [0.1, 0.1, 0.15])
    max_points_per_voxel: int = 5
    max_voxels: int = 1000000
    num_point_features: int = 5
    num_filters: List[int] = field(default_factory=lambda: [16, 32, 64, 128])      # BACKBONE_3D.NUM_FILTERS
    out_features: int = 128                                                        # BACKBONE_3D.OUT_FEATURES
    bev_layer_nums: List[int] = field(default_factory=lambda: [5, 5])              # BACKBONE_2D.*
    bev_layer_strides: List[int] = field(default_factory=lambda: [1, 2])
    bev_num_filters: List[int] = field(default_factory=lambda: [128, 256])
    bev_upsample_strides: List[int] = field(default_factory=lambda: [1, 2])
    bev_num_upsample_filters: List[int] = field(default_factory=lambda: [256, 256])
    shared_conv_channel: int = 64                                                  # DENSE_HEAD.*
    num_class: int = 3
    head_order: List[str] = field(default_factory=lambda: ["center", "center_z", "dim", "rot"])
    head_channels: Dict[str, int] = field(default_factory=lambda: {"center": 2, "center_z": 1, "dim": 3, "rot": 2})
    feature_map_stride: int = 8
    score_thresh: float = 0.1
    post_center_limit_range: List[float] = field(default_factory=lambda: [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0])
    max_obj_per_sample: int = 500
    nms_thresh: float = 0.8
    nms_pre_maxsize: int = 4096
    nms_post_maxsize: int = 500
    # arithmetic of the convolutions with >= 32 input channels (include/cpd_hip.h): "f16x2" = fp32 operands written as two fp16
    # terms, three partial products accumulated in fp32 on the 16-bit matrix pipe (fp32-level error; fp16's range is guarded per
    # layer, `range_guard` below: any activation magnitude gives the fp32 answer, CPD_GC_F16X2); "bf16x3" = three bf16 terms, six partial
    # products (exact split over the whole fp32 range, CPD_GC_BF16X3); "f32" = fp32-input MFMA everywhere (bitwise an fmaf chain)
    conv_math: str = "f16x2"
    # internal row order of the strided levels: "taps" = every chunk of `row_order_chunk` canonical rows sorted by neighbour
    # pattern, so that the conv kernels' 16-row tap skipping is nearly exact (ops.order_rows_by_taps; a level's exported
    # (features, indices) pair is in that order -- any order is a valid sparse tensor); "canonical" = ascending (b, z, y, x)
    # rows of level 0 inside the engine: "canonical" = ascending (frame, z, y, x), straight from the occupancy bitmap's ranks
    # (no first-appearance scan, canonical level-0 index, coherent lookups for the rulebooks built on it); "appearance" = the
    # voxelizer's boundary order (Point2VoxelCPU3d). Same voxels and features either way.
    # f16x2 only: every conv epilogue records the bits of max |output| and the next layer pre-scales its input by a power of
    # two taken from it (exact; the `*_f16s_*` kernels), so that an activation of ANY fp32 magnitude gives the fp32 answer where
    # unguarded split-fp16 arithmetic overflows to inf / NaN at 65504. False = the unscaled kernels.
    range_guard: bool = True
    voxel_row_order: str = "canonical"
    # storage of the activations BETWEEN the f16x2 sparse layers of levels 2-4 (engine-internal; every exported tensor is fp32):
    # True = fp16-pair rows (CPD_GC_*_PAIRS: the epilogue that produces a row writes its split x = h + l, the up to 27 gathers of
    # the row take the bits as MFMA fragments -- same products, no split in the stage loop); a range-guard re-run uses fp32 rows
    pair_rows: bool = True
    pair_rows_level1: bool = True          # ... and level 1 (16 channels: K = 16 MFMA form of the wave kernel) as well
    # ... and the DENSE half (round 5): conv_out writes pair rows, densify copies them, every BEV / head map between two split-fp16 layers
    # is a pair-row map (window_conv_f16p_kernel / tile_conv_f16p_kernel take the stored bits as MFMA fragments: no split in the
    # staging); the head's output maps are fp32. Only batches whose layers all take a window tile the pair kernels exist for (>= 2
    # frames at 188 x 188; engine.dense_pairs_ok): smaller batches keep fp32 dense maps. Exported tensors (bev_cat, encoded) are fp32.
    pair_rows_dense: bool = True
    persistent_dense_map: bool = True      # densify into a persistent pre-zeroed map, re-zero the occupied rows after its reader (ops.DenseMap)
    voxelizer_group: int = 64             # frames per batched-voxelizer call (its FrameOffsets kernel argument holds 64)
    batched_voxelizer_min_frames: int = 2  # smaller batches: one voxelizer call per frame + a separate level-0 index build
    # "bricks" (round 4, measured, NOT the default): rows of the strided levels in 8 x 8 (y, x) brick order of their z-plane,
    # pattern-sorted inside every 128-row tile (ops.order_rows_bricks); with plan_rulebooks their sub-manifold rulebooks are PLANNED
    # (ops.rulebook_plan) and the SparseBasicBlock convs of levels 2-4 run the staged row-wave kernel, which fetches a tile's distinct
    # input rows once into LDS instead of gathering every (row, tap) pair through the vector L1 (cpd_gather_conv_planned; needs
    # pair_rows). Same box, 48 frames: taps 1057, bricks 1045 (rulebooks -9 us/frame, row-wave kernels +5 %: coarser tap skipping),
    # bricks + plan 956 (the staged kernel moves half the bytes through the L1 but holds 8-12 waves per CU, not 25).
    row_order: str = "taps"
    row_order_brick: tuple = (8, 8)
    # "taps": level 0 (the voxel list) in tap-pattern order as well, when the voxelizer delivered canonical rows. Measured, same box, two
    # pairs: level-1 convs 51.8 -> 46.7 us/frame, rulebooks + the sort + the feature gather +6 us: 1046 vs 1049.5 frames/s -> off
    row_order_level0: bool = False
    chunked_rulebooks: bool = True         # "taps": rulebooks of the re-ordered levels built chunk-wise in canonical order (ops.rulebook_*(canonical=))
    plan_channels: tuple = (32, 64)        # levels (by channel count) whose SubM convs take the staged kernel when plan_rulebooks
    plan_tile_rows: int = 256              # rows per workgroup of the staged kernel / per pattern-sorted tile of the brick order (128 | 256)
    plan_rulebooks: bool = False           # "bricks" only: plan the sub-manifold rulebooks -> the staged row-wave kernel
    row_order_chunk: int = 4096
    row_order_min_rows: int = 65536        # below this a level does not fill the chip either way
    # the strided stages' index chain (output set, row order, rulebooks) on a second HIP stream, one stage ahead of the convolutions
    # (engine.backbone3d); batches above `index_side_stream_max_frames` fill the chip with every launch and keep one stream
    index_side_stream: bool = True
    # ... and the BEV deblocks that have a level of convolutions between them and the shared conv, for batches of up to
    # `deblock_side_stream_max_frames` frames (same box: one frame 2.84 -> 2.79 ms, 4 frames +0.6 %; at 48 frames the HBM-bound GEMM
    # slows the window convs it runs beside by more than it hides: 1240 -> 1227 frames/s)
    deblock_side_stream: bool = True
    deblock_side_stream_max_frames: int = 8
    index_side_stream_max_frames: int = 1 << 30

    @property
    def grid_zyx(self):
        return ops.voxel_grid_size(self.voxel_size, self.point_cloud_range)

    @property
    def sparse_shape(self):
        g = self.grid_zyx
        return [g[0] + 1, g[1], g[2]]     # grid_size[::-1] + [1,0,0]  (spconv_backbone.py:412)

    def head_names(self):
        return list(self.head_order) + ["hm"]

    def head_out(self, name):
        return self.num_class if name == "hm" else self.head_channels[name]


# Sparse stages of VoxelResBackBone8x (spconv_backbone.py:414-455): (name, ksize, stride, pad)
_DOWN = {
    "conv2": ([3, 3, 3], [2, 2, 2], [1, 1, 1]),
    "conv3": ([3, 3, 3], [2, 2, 2], [1, 1, 1]),
    "conv4": ([3, 3, 3], [2, 2, 2], [0, 1, 1]),
    "conv_out": ([3, 1, 1], [2, 1, 1], [0, 0, 0]),
}


def init_state_dict(cfg: ModelConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights under the REFERENCE's state_dict names and layouts (spconv-2.x conv
    weights (Cout,kD,kH,kW,Cin); torch Conv2d / ConvTranspose2d / BatchNorm). BN statistics are
    non-trivial so that folding errors would show up in parity tests."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv3d(name, cin, cout, k, bias):
        fan = cin * k[0] * k[1] * k[2]
        sd[name + ".weight"] = torch.randn(cout, k[0], k[1], k[2], cin, generator=g) * math.sqrt(2.0 / fan)
        if bias:
            sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.05

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=g) * 0.5 + 0.75
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_var"] = torch.rand(c, generator=g) * 0.5 + 0.75
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def conv2d(name, cin, cout, k, bias, transposed=False):
        fan = cin * k * k
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        sd[name + ".weight"] = torch.randn(*shape, generator=g) * math.sqrt(2.0 / fan)
        if bias:
            sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.05

    nf = cfg.num_filters
    p = "backbone_3d."
    conv3d(p + "conv_input.0", cfg.num_point_features, nf[0], [3, 3, 3], False)
    bn(p + "conv_input.1", nf[0])

    def basic_block(name, c):
        conv3d(name + ".conv1", c, c, [3, 3, 3], True)
        bn(name + ".bn1", c)
        conv3d(name + ".conv2", c, c, [3, 3, 3], True)
        bn(name + ".bn2", c)

    basic_block(p + "conv1.0", nf[0])
    basic_block(p + "conv1.1", nf[0])
    for lvl, stage in enumerate(["conv2", "conv3", "conv4"], start=1):
        k = _DOWN[stage][0]
        conv3d(p + stage + ".0.0", nf[lvl - 1], nf[lvl], k, False)
        bn(p + stage + ".0.1", nf[lvl])
        basic_block(p + stage + ".1", nf[lvl])
        basic_block(p + stage + ".2", nf[lvl])
    conv3d(p + "conv_out.0", nf[3], cfg.out_features, _DOWN["conv_out"][0], False)
    bn(p + "conv_out.1", cfg.out_features)

    # BaseBEVBackbone (base_bev_backbone.py:27-59)
    p = "backbone_2d."
    depth = 2  # sparse_shape z after conv_out for a 41-deep input; generic value computed by engine
    c_in_list = [cfg.out_features * depth] + cfg.bev_num_filters[:-1]
    for lvl in range(len(cfg.bev_layer_nums)):
        c = cfg.bev_num_filters[lvl]
        conv2d(p + "blocks.%d.1" % lvl, c_in_list[lvl], c, 3, False)
        bn(p + "blocks.%d.2" % lvl, c)
        for k in range(cfg.bev_layer_nums[lvl]):
            conv2d(p + "blocks.%d.%d" % (lvl, 4 + 3 * k), c, c, 3, False)
            bn(p + "blocks.%d.%d" % (lvl, 5 + 3 * k), c)
        u = cfg.bev_upsample_strides[lvl]
        conv2d(p + "deblocks.%d.0" % lvl, c, cfg.bev_num_upsample_filters[lvl], u, False, transposed=True)
        bn(p + "deblocks.%d.1" % lvl, cfg.bev_num_upsample_filters[lvl])

    # CenterHead (center_head.py:73-94, SeparateHead l.11-45; USE_BIAS_BEFORE_NORM True)
    p = "dense_head."
    c_cat = sum(cfg.bev_num_upsample_filters)
    sc = cfg.shared_conv_channel
    conv2d(p + "shared_conv.0", c_cat, sc, 3, True)
    bn(p + "shared_conv.1", sc)
    for name in cfg.head_names():
        q = p + "heads_list.0.%s." % name
        conv2d(q + "0.0", sc, sc, 3, True)
        bn(q + "0.1", sc)
        conv2d(q + "1", sc, cfg.head_out(name), 3, True)
        if name == "hm":
            sd[q + "1.bias"] = torch.full((cfg.num_class,), -2.19)   # center_head.py:30
    return sd


def _fold_bn(sd, bn_name, eps, conv_bias=None):
    """Eval BatchNorm (+ preceding conv bias) as per-channel scale/shift."""
    w, b = sd[bn_name + ".weight"].double(), sd[bn_name + ".bias"].double()
    mean, var = sd[bn_name + ".running_mean"].double(), sd[bn_name + ".running_var"].double()
    scale = w / torch.sqrt(var + eps)
    shift = b - mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias.double() * scale
    return scale.float(), shift.float()


class _Layer:
    __slots__ = ("w", "kv", "c_in", "c_out", "scale", "shift", "relu")

    def __init__(self, w_kio, scale, shift, relu, device):
        w_kio = w_kio.to(device=device, dtype=torch.float32).contiguous()
        self.kv, self.c_in, self.c_out = w_kio.shape
        self.w = ops.pack_weight(w_kio)
        self.scale = scale.to(device).contiguous() if scale is not None else None
        self.shift = shift.to(device).contiguous() if shift is not None else None
        self.relu = relu


class CenterPointEngine:
    """points [N, C] (device f32)  ->  {'pred_boxes','pred_scores','pred_labels'} per frame."""

    SPARSE_BN_EPS = 1e-3   # spconv_backbone.py:410
    BEV_BN_EPS = 1e-3      # base_bev_backbone.py:38
    HEAD_BN_EPS = 1e-5     # nn.BatchNorm2d default, center_head.py:24,78

    def __init__(self, cfg: ModelConfig, state_dict: Dict[str, torch.Tensor], device="cuda", host_results=False):
        self.cfg = cfg
        self.device = torch.device(device)
        # host_results: the final boxes / scores / labels are returned as HOST tensors (the D2H of the padded result block
        # follows the step's one count read-back)
        self.host_results = bool(host_results)
        self.sd = state_dict
        self.voxelizer = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, cfg.num_point_features,
                                       cfg.max_points_per_voxel, cfg.max_voxels, device=self.device)
        self._voxelizers = [self.voxelizer]
        self._group_voxelizers = []
        self._build_sparse()
        self._bev_cache = {}
        self._build_dense()

    # ------------------------------------------------------------------ weights
    def _sparse_w(self, name):
        w = self.sd[name + ".weight"]                     # (Cout, kD, kH, kW, Cin)
        cout, cin = w.shape[0], w.shape[-1]
        return w.reshape(cout, -1, cin).permute(1, 2, 0)   # [kv, Cin, Cout]

    def _build_sparse(self):
        sd, dev, eps = self.sd, self.device, self.SPARSE_BN_EPS
        p = "backbone_3d."
        L = {}
        s, t = _fold_bn(sd, p + "conv_input.1", eps)
        L["conv_input"] = _Layer(self._sparse_w(p + "conv_input.0"), s, t, True, dev)

        def block(name):
            s1, t1 = _fold_bn(sd, name + ".bn1", eps, sd.get(name + ".conv1.bias"))
            s2, t2 = _fold_bn(sd, name + ".bn2", eps, sd.get(name + ".conv2.bias"))
            return (_Layer(self._sparse_w(name + ".conv1"), s1, t1, True, dev),
                    _Layer(self._sparse_w(name + ".conv2"), s2, t2, True, dev))   # relu after the residual add

        L["conv1"] = [block(p + "conv1.0"), block(p + "conv1.1")]
        for stage in ["conv2", "conv3", "conv4"]:
            s, t = _fold_bn(sd, p + stage + ".0.1", eps)
            L[stage + ".down"] = _Layer(self._sparse_w(p + stage + ".0.0"), s, t, True, dev)
            L[stage] = [block(p + stage + ".1"), block(p + stage + ".2")]
        s, t = _fold_bn(sd, p + "conv_out.1", eps)
        L["conv_out"] = _Layer(self._sparse_w(p + "conv_out.0"), s, t, True, dev)
        self.sparse = L

    def _build_dense(self):
        cfg, sd, dev = self.cfg, self.sd, self.device
        p = "backbone_2d."
        depth = self._final_depth()
        C = cfg.out_features
        self.bev_levels = []
        for lvl in range(len(cfg.bev_layer_nums)):
            convs = []
            names = ["blocks.%d.1" % lvl] + ["blocks.%d.%d" % (lvl, 4 + 3 * k) for k in range(cfg.bev_layer_nums[lvl])]
            bns = ["blocks.%d.2" % lvl] + ["blocks.%d.%d" % (lvl, 5 + 3 * k) for k in range(cfg.bev_layer_nums[lvl])]
            for i, (cn, bnn) in enumerate(zip(names, bns)):
                w = sd[p + cn + ".weight"]                                   # (Cout, Cin, 3, 3)
                if lvl == 0 and i == 0:
                    # reference channel = c*D + z (height_compression.py:136-138); ours = z*C + c
                    cout = w.shape[0]
                    w = w.reshape(cout, C, depth, 3, 3).permute(0, 2, 1, 3, 4).reshape(cout, depth * C, 3, 3)
                s, t = _fold_bn(sd, p + bnn, self.BEV_BN_EPS)
                convs.append(_Layer(w.permute(2, 3, 1, 0).reshape(9, w.shape[1], w.shape[0]), s, t, True, dev))
            u = cfg.bev_upsample_strides[lvl]
            wd = sd[p + "deblocks.%d.0.weight" % lvl]                        # (Cin, Cout, u, u)
            s, t = _fold_bn(sd, p + "deblocks.%d.1" % lvl, self.BEV_BN_EPS)
            cin, cout = wd.shape[0], wd.shape[1]
            # ConvTranspose2d(k=s=u) = one 1x1 GEMM with the u*u taps stacked along the columns
            w_kio = wd.permute(0, 2, 3, 1).reshape(1, cin, u * u * cout)
            de = _Layer(w_kio, s.repeat(u * u), t.repeat(u * u), True, dev)
            self.bev_levels.append((convs, de, u, cout))
        self._build_head()

    def _build_head(self):
        """CenterHead's convolutions (center_head.py:73-94); a subclass with another dense head (cpd_amd/anchor_engine.py) overrides this,
        `_head_rows` and `decode_and_nms`."""
        cfg, sd, dev = self.cfg, self.sd, self.device
        p = "dense_head."
        s, t = _fold_bn(sd, p + "shared_conv.1", self.HEAD_BN_EPS, sd.get(p + "shared_conv.0.bias"))
        w = sd[p + "shared_conv.0.weight"]
        self.shared = _Layer(w.permute(2, 3, 1, 0).reshape(9, w.shape[1], w.shape[0]), s, t, True, dev)
        # the five SeparateHead branches: first convs fused along Cout, second convs block-diagonal
        names = cfg.head_names()
        sc = cfg.shared_conv_channel
        w1, s1, t1 = [], [], []
        n_out = sum(cfg.head_out(n) for n in names)
        w2 = torch.zeros(9, sc * len(names), n_out)
        b2 = torch.zeros(n_out)
        self.head_slices = {}
        col = 0
        for hi, name in enumerate(names):
            q = p + "heads_list.0.%s." % name
            w = sd[q + "0.0.weight"]
            w1.append(w.permute(2, 3, 1, 0).reshape(9, sc, sc))
            s, t = _fold_bn(sd, q + "0.1", self.HEAD_BN_EPS, sd.get(q + "0.0.bias"))
            s1.append(s); t1.append(t)
            wo = sd[q + "1.weight"]                                          # (co, sc, 3, 3)
            co = wo.shape[0]
            w2[:, hi * sc:(hi + 1) * sc, col:col + co] = wo.permute(2, 3, 1, 0).reshape(9, sc, co)
            b2[col:col + co] = sd[q + "1.bias"]
            self.head_slices[name] = (col, co)
            col += co
        self.head1 = _Layer(torch.cat(w1, dim=2), torch.cat(s1), torch.cat(t1), True, dev)
        self.head2 = _Layer(w2, None, b2, False, dev)
        self.head_ld = 16 * ((n_out + 15) // 16)

    def _final_shape(self):
        shape = self.cfg.sparse_shape
        for stage in ["conv2", "conv3", "conv4", "conv_out"]:
            k, s, pd = _DOWN[stage]
            shape = ops.conv_out_shape(shape, k, s, pd)
        return shape

    def _final_depth(self):
        return self._final_shape()[0]

    def _side_stream(self):
        return _lib.side_stream(self.device, "index")      # (per current stream: an engine may be driven from several streams over its life)

    # ------------------------------------------------------------------ forward pieces
    # Range guard of the f16x2 path. Every conv epilogue raises an absmax block to max |out| (a wave
    # reduction + a rarely issued atomic: free). OPTIMISTIC pass: the layers run their unscaled kernels and only RECORD; the largest
    # recorded value rides along with the step's final count read-back, and if any activation reached 2^15 the step is run again
    # in the GUARDED mode, where each layer takes its input's block as `in_absmax` (the `*_f16s_*` kernels pre-scale by a power of
    # two: exact at any magnitude). Always guarded costs the dense kernels ~3 % (the extra multiply in their staging); a re-run
    # costs a step, once, for inputs the unguarded arithmetic would have turned into inf / NaN (or, behind a ReLU, into zeros).
    # `self._rb` = block of the tensor the NEXT conv reads; a _conv call consumes it and replaces it by the block its epilogue fills.
    def _range_reset(self):
        self._rb = None
        self._rb_next = 0
        if self.cfg.conv_math == "f16x2" and self.cfg.range_guard:
            if getattr(self, "_rb_pool", None) is None:
                # one block per conv launch of a step: 20 sparse layers + conv_out, the BEV levels' convs, the concat buffer,
                # shared conv + the two head launches (+ slack for callers that run the stages on their own)
                n = 21 + sum(k + 1 for k in self.cfg.bev_layer_nums) + 1 + 3 + 16
                self._rb_pool = ops.absmax_blocks(n, self.device)
            else:
                self._rb_pool.zero_()
        else:
            self._rb_pool = None

    def _range_new(self):
        if self._rb_pool is None:
            return None
        if self._rb_next >= self._rb_pool.shape[0]:
            raise RuntimeError("range guard: this model launches more convolutions per step than the %d absmax blocks the engine "
                               "sized from its config (engine._range_reset)" % self._rb_pool.shape[0])
        b = self._rb_pool[self._rb_next]
        self._rb_next += 1
        return b

    RANGE_STICKY_STEPS = 16       # guarded steps that follow a re-run before the engine tries the unguarded kernels again

    def _range_exceeded_flag(self):
        """device int32 [1] riding with the step's count read-back; None when nothing is recorded. Optimistic step: 1 when some
        recorded activation is >= 2^15 (or NaN / inf) -> the step is re-run guarded. Guarded step: 1 while some activation is
        still >= 2^14 -> the engine stays guarded (hysteresis: it returns to the unguarded kernels only after
        RANGE_STICKY_STEPS consecutive steps well inside fp16's range)."""
        if self._rb_pool is None:
            return None
        thr = 0x46800000 if getattr(self, "_rb_scaled", False) else 0x47000000   # bits of 16384.0f / 32768.0f; NaN / inf bits are larger
        return (self._rb_pool.max() >= thr).to(torch.int32).view(1)

    def _conv(self, layer, x, nbr, n_out, residual=None, out=None, out_row_map=None, out_col_group=0, dense=False, out_rb="new",
              in_pairs=False, out_pairs=False, res_pairs=False):
        rb_in = self._rb if getattr(self, "_rb_scaled", False) else None
        rb_out = self._range_new() if out_rb == "new" else out_rb
        y = ops.gather_conv(x, layer.c_in, layer.w, nbr, layer.kv, n_out, layer.c_out, layer.scale, layer.shift,
                            residual, layer.relu, out=out, out_row_map=out_row_map, out_col_group=out_col_group,
                            dense=dense, math=self.cfg.conv_math, in_absmax=rb_in, out_absmax=rb_out,
                            in_pairs=in_pairs, out_pairs=out_pairs, res_pairs=res_pairs)
        self._rb = rb_out
        return y

    def _blocks(self, blocks, x, nbr, pairs=False):
        n = x.shape[0]
        for c1, c2 in blocks:                      # SparseBasicBlock, spconv_backbone.py:120-136
            y = self._conv(c1, x, nbr, n, in_pairs=pairs, out_pairs=pairs)
            x = self._conv(c2, y, nbr, n, residual=x, in_pairs=pairs, out_pairs=pairs, res_pairs=pairs)
        return x

    def backbone3d(self, feats, coords, batch, index=None, pair_rows=False, export_levels=True, canonical0=False, dense_pairs=False):
        """VoxelResBackBone8x.forward (spconv_backbone.py:502-558). Returns per-level
        {name: (features, indices, spatial_shape)} and the stride-8 output. `index`: the level-0 site index when the
        voxelizer already built it (cpd_voxelize_batch_index). `pair_rows`: levels 2-4 keep their activations as fp16-pair rows
        between layers (ModelConfig.pair_rows; f16x2 only); the returned level features are fp32 either way when
        `export_levels` (decoded copies), else whatever the layers left."""
        pairs = bool(pair_rows) and self.cfg.conv_math == "f16x2"
        want = (lambda name: name in export_levels) if isinstance(export_levels, (tuple, list, set)) else (lambda name: bool(export_levels))
        self.level_indexes = {}                               # name -> SiteIndex of the level (in the level's row order)
        L = self.sparse
        shape = self.cfg.sparse_shape
        # The INDEX CHAIN of the strided stages (output set -> row order -> rulebooks: 6-8 small launches and one count read-back per
        # stage) depends on site lists only, never on features. With `index_side_stream` it runs on a second HIP stream, ahead of the
        # convolutions, and PIPELINED: a stage's output set is marked and counted (conv_outset_begin) as soon as the list it is marked
        # from exists -- stage 2's right here, before the level-0 index, rulebook and the level-1 convs are queued; stage k + 1's as
        # soon as stage k's list is emitted -- so that by the time the host asks for a count (conv_outset_end) the side stream has it,
        # and the host is never away from the main stream for long. The index kernels overlap convolutions that do not fill the chip
        # (one frame: 0.6 of 3.0 ms of kernel time is index work). Same kernels, same results.
        side = self._side_stream() if (self.cfg.index_side_stream and feats.is_cuda and batch <= self.cfg.index_side_stream_max_frames) else None
        main = torch.cuda.current_stream(self.device) if side is not None else None
        stages = ["conv2", "conv3", "conv4", "conv_out"]
        self._index_keep = []

        def begin(stage, coords_c, shape):
            k, s, pd = _DOWN[stage]
            return ops.conv_outset_begin(coords_c, batch, shape, k, s, pd)

        if side is not None:
            # side-stream tensors read on `main` (tables, lists, indexes) live until the NEXT step's chain starts, and that start waits
            # for everything queued on `main` before this point: the caching allocator may hand a freed block to the side stream only
            # after its readers ran
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                pending = begin(stages[0], coords, shape)
        if index is None:
            index = ops.SiteIndex.build(coords, batch, shape)
        coords_c0, canon0 = coords, None
        if canonical0 and self.cfg.row_order == "taps" and self.cfg.row_order_level0 and coords.shape[0] >= self.cfg.row_order_min_rows:
            # level 0 in tap-pattern order too (round 4): its rows have 4.3 of 27 neighbours on average, and a canonical 16-row group
            # executes 2.2x the (group, tap) pairs its rows need -- 1.3x after the sort (tools/unique_probe.py). `canonical0`: the
            # voxelizer delivered canonical rows and a canonical index (rank = row), which is what the sort and the map need.
            coords, n2o, o2n = ops.order_rows_by_taps(coords_c0, index, chunk_rows=self.cfg.row_order_chunk)
            index.set_order(o2n)
            feats = feats.index_select(0, n2o.long())
            if self.cfg.chunked_rulebooks:
                canon0 = (coords_c0, o2n, self.cfg.row_order_chunk)
        pairs16 = pairs and self.cfg.pair_rows_level1

        def tables(stage, index, pending, after=None):
            """everything of a strided stage that depends on the site lists only: its output set (`pending`: its conv_outset_begin),
            row order, both rulebooks -- and the next stage's conv_outset_begin; index = the site index of the stage's input level,
            `after`: an event of the main stream that index is complete at"""
            k, s, pd = _DOWN[stage]
            out_idx, out_index, out_shape = ops.conv_outset_end(pending)
            if after is not None:
                torch.cuda.current_stream(self.device).wait_event(after)
            if stage == "conv_out":
                return dict(out_idx=out_idx, out_c=out_idx, out_index=out_index, out_shape=out_shape,
                            nbr_dn=ops.rulebook_conv(out_idx, index, k, s, pd), nbr=None, pairs_out=False, pending=None)
            out_c = out_idx
            pending = begin(stages[stages.index(stage) + 1], out_c, out_shape)     # (the canonical list: what the next output set is marked from)
            # "bricks" per level: only where the staged kernel is used (plan_channels: the widths it wins at); other levels keep "taps"
            c_lvl = L[stage + ".down"].c_out
            use_plan = self.cfg.plan_rulebooks and pairs and c_lvl in self.cfg.plan_channels
            bricks = self.cfg.row_order == "bricks" and out_idx.shape[0] >= self.cfg.row_order_min_rows and (use_plan or not self.cfg.plan_rulebooks)
            canon = None                      # (canonical list, order, chunk) of a chunk-ordered level: its tables are built chunk-wise
            if bricks:
                out_idx, _, old_to_new = ops.order_rows_bricks(out_c, out_index, brick=self.cfg.row_order_brick, tile_rows=self.cfg.plan_tile_rows)
                out_index.set_order(old_to_new)
            elif self.cfg.row_order in ("taps", "bricks") and out_idx.shape[0] >= self.cfg.row_order_min_rows:
                # rows of the level sorted, chunk by chunk, by their neighbour pattern (ops.order_rows_by_taps): the level's
                # site list in the new order + the rank -> row map installed in its index re-order everything that follows
                # (both rulebooks, the features, the exported level) without any kernel knowing
                out_idx, _, old_to_new = ops.order_rows_by_taps(out_c, out_index, chunk_rows=self.cfg.row_order_chunk)
                out_index.set_order(old_to_new)
                if self.cfg.chunked_rulebooks:
                    canon = (out_c, old_to_new, self.cfg.row_order_chunk)
            nbr_dn = ops.rulebook_conv(out_idx, index, k, s, pd, canonical=canon)
            # (pair rows are read through the row-wave kernel's 4 GB buffer resource: a level that large stays fp32)
            pairs_out = pairs and out_idx.shape[0] * c_lvl * 4 < 0xfffff000
            nbr = ops.rulebook_subm(out_idx, out_index, canonical=canon)
            if bricks and pairs_out and use_plan:
                ops.rulebook_plan(nbr, self.cfg.plan_tile_rows)  # the level's four SubM convs: the staged row-wave kernel
            return dict(out_idx=out_idx, out_c=out_c, out_index=out_index, out_shape=out_shape, nbr_dn=nbr_dn, nbr=nbr, pairs_out=pairs_out, canon=canon,
                        pending=pending)

        def stage_tables(stage, index, pending, after=None):
            """queue a stage's index chain (on the side stream when there is one)"""
            if side is None:
                T = tables(stage, index, pending)
                T["ev"] = None
            else:
                with torch.cuda.stream(side):
                    T = tables(stage, index, pending, after)
                    T["ev"] = side.record_event()
            self._index_keep.append(T)
            return T

        ev_index = main.record_event() if side is not None else None          # the level-0 index (and its order) are complete here
        nbr = ops.rulebook_subm(coords, index, canonical=canon0)               # 'subm1' and 'res1' are the same L0 table
        self._range_reset()                                  # (the 5-channel input layer runs on the fp32 pipe: no block for `feats`)
        # level 1 (16 channels): with pair rows its layers run the K = 16 split-fp16 MFMA on 16-channel pair rows (three products of
        # 16 matrix cycles instead of four fp32 MFMAs of 32); the 5-channel input layer stays on the fp32 pipe and writes the first pairs
        x = self._conv(L["conv_input"], feats, nbr, coords.shape[0], out_pairs=pairs16)
        x = self._blocks(L["conv1"], x, nbr, pairs=pairs16)
        raw = getattr(self, "_export_pair_levels", False)     # exported levels stay fp16-pair rows, wrapped in ops.PairRows (the RoI pooling's first GEMM reads them as they are)

        def export(t, is_pairs, name):
            if not (is_pairs and want(name)):
                return t
            if raw and t.shape[1] % 32 == 0:
                return ops.PairRows(t)
            return ops.pairs_to_rows(t)
        levels = {"x_conv1": (export(x, pairs16, "x_conv1"), coords, shape)}
        self.level_indexes["x_conv1"] = index
        pairs_in = pairs16                 # (what conv2.down reads)
        if side is None:
            pending = begin(stages[0], coords_c0, shape)
        # (a stage's chain is queued AFTER the previous stage's convs: its count read-back can still block the host for a moment, and
        # the main stream should have its work by then)
        T = stage_tables(stages[0], index, pending, after=ev_index)
        for i, stage in enumerate(stages, start=2):
            if T["ev"] is not None:
                main.wait_event(T["ev"])
            out_idx, out_shape = T["out_idx"], T["out_shape"]
            if stage == "conv_out":
                break
            pairs_out = T["pairs_out"]
            x = self._conv(L[stage + ".down"], x, T["nbr_dn"], out_idx.shape[0], in_pairs=pairs_in, out_pairs=pairs_out)
            x = self._blocks(L[stage], x, T["nbr"], pairs=pairs_out)
            pairs_in = pairs_out
            levels["x_conv%d" % i] = (export(x, pairs_in, "x_conv%d" % i), out_idx, out_shape)
            self.level_indexes["x_conv%d" % i] = T["out_index"]
            T = stage_tables(stages[i - 1], T["out_index"], T["pending"])
        # (dense_pairs: the stride-8 output stays in pair rows -- densify is a copy of row bytes, so the BEV map it builds is a pair-row map)
        x = self._conv(L["conv_out"], x, T["nbr_dn"], out_idx.shape[0], in_pairs=pairs_in, out_pairs=bool(dense_pairs) and pairs_in)
        self.encoded_pairs = bool(dense_pairs) and pairs_in
        self._rb_stage = "backbone"                            # the densified map inherits this output's range block
        return levels, (x, out_idx, out_shape)

    def _bev_tables(self, batch, h, w):
        key = (batch, h, w)
        if key not in self._bev_cache:
            dev = self.device
            t = {}
            t["s1"] = ops.rulebook_conv2d(batch, h, w, 3, 3, 1, 1, dev)
            t["s2"] = ops.rulebook_conv2d(batch, h, w, 3, 3, 2, 1, dev)
            h2, w2 = t["s2"][1], t["s2"][2]
            t["s1_half"] = ops.rulebook_conv2d(batch, h2, w2, 3, 3, 1, 1, dev)
            # ConvTranspose2d(k=s=2) destination rows: tap (a,b) of coarse pixel (y,x) -> (2y+a, 2x+b)
            b_i = torch.arange(batch, device=dev).view(-1, 1, 1)
            yy = torch.arange(h2, device=dev).view(1, -1, 1)
            xx = torch.arange(w2, device=dev).view(1, 1, -1)
            maps = []
            for a in range(2):
                for bb in range(2):
                    maps.append(((b_i * h + 2 * yy + a) * w + 2 * xx + bb).reshape(-1))
            t["up2"] = torch.stack(maps).to(torch.int32).contiguous()
            self._bev_cache[key] = t
        return self._bev_cache[key]

    def dense_pairs_ok(self, batch, h, w):
        """can this batch run its dense half on fp16-pair maps? Every 3 x 3 / stride 1 layer must take one of the window tiles the pair
        kernels exist for (128 x 128, 256 x 64, 256 x 16: cpd_conv3x3_rows_tile), the strided / deblock GEMMs need 128-column tiles,
        and every map must stay below the 4 GB a buffer resource addresses."""
        cfg = self.cfg
        if not (cfg.pair_rows_dense and cfg.pair_rows and cfg.conv_math == "f16x2"):
            return False
        key = ("dense_pairs", batch, h, w)
        if key in self._bev_cache:
            return self._bev_cache[key]
        ok = True
        cur_h, cur_w = h, w
        widest = self.bev_levels[0][0][0].c_in
        for lvl, (convs, de, u, c_up) in enumerate(self.bev_levels):
            stride = cfg.bev_layer_strides[lvl]
            if stride == 2:
                cur_h, cur_w = (cur_h + 2 - 3) // 2 + 1, (cur_w + 2 - 3) // 2 + 1
                ok = ok and convs[0].c_out % 128 == 0 and convs[0].c_in % 32 == 0
            for cv in (convs if stride == 1 else convs[1:]):
                ok = ok and ops.conv3x3_rows_tile(batch, cur_h, cur_w, cv.c_in, cv.c_out) in ((128, 128), (128, 64))
            ok = ok and de.c_out % 128 == 0 and c_up % 32 == 0
            widest = max(widest, max(cv.c_out for cv in convs))
        c_cat = sum(cfg.bev_num_upsample_filters)
        ok = ok and self._head_pairs_ok(batch, h, w, c_cat)
        widest = max(widest, c_cat, getattr(getattr(self, "head1", None), "c_out", 0))
        ok = ok and batch * h * w * widest * 4 + (w + 300) * widest * 4 < 0xfff00000
        self._bev_cache[key] = bool(ok)
        return bool(ok)

    def _head_pairs_ok(self, batch, h, w, c_cat):
        return (ops.conv3x3_rows_tile(batch, h, w, c_cat, self.shared.c_out) in ((256, 64), (128, 64)) and
                ops.conv3x3_rows_tile(batch, h, w, self.head1.c_in, self.head1.c_out) in ((256, 64), (128, 64)) and
                ops.conv3x3_rows_tile(batch, h, w, self.head2.c_in, self.head2.c_out) in ((256, 16), (128, 16)))

    def bev_and_head(self, dense_rows, batch, h, w, pairs=False):
        """BaseBEVBackbone.forward (base_bev_backbone.py:85-122) + CenterHead convs
        (center_head.py:323-330) on channels-last rows [batch*h*w, C]. `pairs`: `dense_rows` and every map up to the head's hidden
        layer are fp16-pair maps (ModelConfig.pair_rows_dense); the returned concat map is then a pair map, the head rows are fp32."""
        cfg = self.cfg
        pk = dict(in_pairs=True, out_pairs=True) if pairs else {}
        T = self._bev_tables(batch, h, w)
        if getattr(self, "_rb_stage", None) != "backbone":     # called on its own (tests, tools): the input's range is unknown
            self._range_reset()
        self._rb_stage = None
        cat_rb = self._range_new()                             # ONE block for the concat buffer: both deblocks raise it
        n_full = batch * h * w
        c_cat = sum(cfg.bev_num_upsample_filters)
        cat = torch.empty((n_full, c_cat), dtype=torch.float32, device=self.device)
        x = dense_rows
        col = 0
        cur_h, cur_w = h, w
        side = self._side_stream() if (cfg.index_side_stream and cfg.deblock_side_stream and dense_rows.is_cuda
                                       and batch <= cfg.deblock_side_stream_max_frames) else None
        main = torch.cuda.current_stream(self.device) if side is not None else None
        held, joins = [], []
        for lvl, (convs, de, u, c_up) in enumerate(self.bev_levels):
            stride = cfg.bev_layer_strides[lvl]
            if stride == 1:
                nbr0, ho, wo = T["s1"] if (cur_h, cur_w) == (h, w) else T["s1_half"]
            elif stride == 2 and (cur_h, cur_w) == (h, w):
                nbr0, ho, wo = T["s2"]
            else:
                raise NotImplementedError("BEV stride pattern outside the shipped configs")
            n_lvl = batch * ho * wo
            x = self._conv(convs[0], x, nbr0, n_lvl, dense=True, **pk)
            if lvl == 0 and getattr(self, "_dense_map", None) is not None:
                self._dense_map.clear()                        # the densified map's only reader is queued: its rows go back to zero
                self._dense_map = None
            nbr_same = T["s1"][0] if (ho, wo) == (h, w) else T["s1_half"][0]
            for cv in convs[1:]:
                x = self._conv(cv, x, nbr_same, n_lvl, dense=True, **pk)
            cur_h, cur_w = ho, wo
            dst = cat[:, col:col + c_up]
            x_rb = self._rb                                    # the next level goes on from x, not from the deblock's output
            # a deblock that is not the last one has a whole level of convolutions between it and its reader (the shared conv): with the
            # side stream it runs beside them -- an HBM-bound GEMM (2.6 GB per 48 frames) under matrix-bound window convs
            aside = side is not None and lvl + 1 < len(self.bev_levels)
            if aside:
                ev = main.record_event()
                held.append(x)                                 # (main-stream tensor read on the side stream: alive until the join below)
            with (torch.cuda.stream(side) if aside else contextlib.nullcontext()):
                if aside:
                    side.wait_event(ev)
                if u == 1:
                    self._conv(de, x, None, n_lvl, out=dst, dense=True, out_rb=cat_rb, **pk)
                elif u == 2 and (ho * 2, wo * 2) == (h, w):
                    self._conv(de, x, None, n_lvl, out=dst, out_row_map=T["up2"], out_col_group=c_up, dense=True, out_rb=cat_rb, **pk)
                else:
                    raise NotImplementedError("upsample stride outside the shipped configs")
                if aside:
                    joins.append(side.record_event())
            self._rb = x_rb
            col += c_up
        for ev in joins:
            main.wait_event(ev)
        self._rb = cat_rb
        return cat, self._head_rows(cat, batch, h, w, T, pairs)

    def _head_rows(self, cat, batch, h, w, T, pairs):
        """the dense head's convolutions on the concat map's rows -> what decode_and_nms takes (CenterHead: the head rows)"""
        pk = dict(in_pairs=True, out_pairs=True) if pairs else {}
        n_full = batch * h * w
        s = self._conv(self.shared, cat, T["s1"][0], n_full, dense=True, **pk)
        h1 = self._conv(self.head1, s, T["s1"][0], n_full, dense=True, **pk)
        out = torch.empty((n_full, self.head_ld), dtype=torch.float32, device=self.device)
        self._conv(self.head2, h1, T["s1"][0], n_full, out=out, dense=pairs, in_pairs=pairs)
        return out

    def decode_and_nms(self, head_rows, batch, h, w, raw=False):
        """generate_predicted_boxes (center_head.py:252-303) + class_agnostic_nms
        (model_nms_utils.py:115-134) for all samples of the batch: five launches, nothing read back
        until the final per-sample counts."""
        cfg = self.cfg
        ld = self.head_ld
        sl = self.head_slices
        base = head_rows
        boxes, scores, labels, counts = ops.center_decode(
            base[:, sl["hm"][0]:], base[:, sl["center"][0]:], base[:, sl["center_z"][0]:], base[:, sl["dim"][0]:],
            base[:, sl["rot"][0]:], ld, 1, cfg.num_class, h, w, cfg.max_obj_per_sample, float(cfg.feature_map_stride),
            cfg.voxel_size[:2], cfg.point_cloud_range[:2], cfg.post_center_limit_range, cfg.score_thresh, sync=False,
            batch=batch, sample_stride=h * w * ld)
        # scores come out sorted descending (top-K order survives the masks), so the
        # topk(NMS_PRE_MAXSIZE) + sort inside nms_gpu are identities (K = 500 <= 4096)
        assert cfg.max_obj_per_sample <= cfg.nms_pre_maxsize
        keep, num_keep = ops.nms_batch(boxes, counts, cfg.nms_thresh)
        flag = self._range_exceeded_flag()        # the range guard's verdict travels with the counts: no extra synchronisation
        # counts, the verdict and the padded boxes / scores / labels are ONE device block: with host results one blocking copy brings a
        # step's whole output over (they were four: the counts, then three tensors at ~25 us of synchronisation each)
        ob, os_, ol, on, blk = ops.select_boxes(boxes, scores, labels, keep, num_keep, cfg.nms_post_maxsize, label_offset=1,
                                                packed=True, extra_ints=1)
        lay = blk._cpd_layout
        hdr = blk[:4 * lay["n_hdr"]].view(torch.int32)
        if flag is not None:
            hdr[batch:batch + 1].copy_(flag)
        whole = self.host_results and not raw
        # (blocking copies. Asynchronous copies into pinned memory queued on the compute stream were measured 5-8 ms per step SLOWER on
        # MI355X / ROCm 7.2 -- tools/d2h_probe.py -- whatever the wait that followed them: event, stream or a later blocking read.)
        host = blk.cpu() if whole else None       # the one host read-back of the stage
        ns = (ops.unpack_boxes(host, lay)[0] if whole else hdr).tolist()
        high = bool(ns[batch]) if flag is not None else False
        if getattr(self, "_rb_scaled", False):
            self._range_exceeded, self._range_high = False, high     # guarded already: exact whatever the range; `high` keeps it guarded
        else:
            self._range_exceeded = high
        ns = ns[:batch]
        if self._range_exceeded:
            return None                           # forward() runs the step again, guarded
        if raw:                                   # the padded device block + per-frame counts (the two-stage engine's proposals)
            return ob, os_, ol, ns
        if whole:
            _, ob, os_, ol = ops.unpack_boxes(host, lay)
        return [{"pred_boxes": ob[b, :ns[b]], "pred_scores": os_[b, :ns[b]], "pred_labels": ol[b, :ns[b]]}
                for b in range(batch)]

    # ------------------------------------------------------------------ whole frame(s)
    @torch.no_grad()
    def forward(self, points_list, return_intermediates=False, proposals=None, pair_levels=False):
        """points_list: list of [N_i, C] device tensors (one per frame of the batch).
        pair_levels (with proposals): the exported levels of >= 32 channels are left as fp16-pair rows (ops.PairRows instead of a tensor)
        when the step ran on pair rows -- what roi_pool's first GEMM takes directly; default fp32 rows.
        proposals = a collection of level names: the first stage of a two-stage detector -- returns
        (boxes [B, cap, 7], scores, labels i64 (1-based), per-frame counts (host list), levels) with the named levels' features
        exported as fp32 rows and their site indexes in self.level_indexes; nothing but the counts leaves the device."""
        if isinstance(points_list, torch.Tensor):
            points_list = [points_list]
        batch = len(points_list)
        index0 = None
        canonical0 = False
        z_extra = self.cfg.sparse_shape[0] - self.cfg.grid_zyx[0]
        if batch >= self.cfg.batched_voxelizer_min_frames and self.voxelizer.batch_supported(batch, z_extra):
            # one set of voxelizer launches for the whole batch; rows come out frame after frame; the voxelizer's occupancy
            # bitmap / prefix / rank -> row map ARE the level-0 site index of the backbone
            n_pts = sum(int(p.shape[0]) for p in points_list)
            canonical = self.cfg.voxel_row_order == "canonical" and batch * self.cfg.max_voxels >= n_pts
            _, coords, _, feats, nvox, index0 = self.voxelizer.batch(points_list, index_z_extra=z_extra, canonical=canonical)
            counts = nvox.tolist()                          # the one read-back: per-frame counts + total
            if canonical and max(counts[:batch]) > self.cfg.max_voxels:
                # a frame above its max_voxels cap: which voxels survive is defined on first-appearance order -- the exact path
                _, coords, _, feats, nvox, index0 = self.voxelizer.batch(points_list, index_z_extra=z_extra)
                counts = nvox.tolist()
            total = counts[batch]
            feats, coords = feats[:total], coords[:total]
            canonical0 = canonical and max(counts[:batch]) <= self.cfg.max_voxels
        elif batch > 1 and self.voxelizer.batch_supported(self.cfg.voxelizer_group, z_extra):
            # more frames than one batched-voxelizer call takes: groups of `voxelizer_group` frames, one voxelizer
            # (workspace) per group, rows concatenated with the frame index offset; the level-0 index is built over the whole list
            g = self.cfg.voxelizer_group
            groups = [points_list[i:i + g] for i in range(0, batch, g)]
            if len(groups[-1]) == 1:                        # e.g. batch = g + 1: the batched call wants >= 2 frames -- borrow one
                groups[-1].insert(0, groups[-2].pop())
            while len(self._group_voxelizers) < len(groups):
                self._group_voxelizers.append(ops.Voxelizer(self.cfg.voxel_size, self.cfg.point_cloud_range, self.cfg.num_point_features,
                                                            self.cfg.max_points_per_voxel, self.cfg.max_voxels, device=self.device))
            outs = [vz.batch(grp) for vz, grp in zip(self._group_voxelizers, groups)]
            totals = torch.stack([o[4][len(grp)] for o, grp in zip(outs, groups)]).tolist()       # the one read-back
            feats = torch.cat([o[3][:m] for o, m in zip(outs, totals)])
            parts, first = [], 0
            for o, m, grp in zip(outs, totals, groups):
                c = o[1][:m].clone()
                c[:, 0] += first                            # frame index within the whole batch
                parts.append(c)
                first += len(grp)
            coords = torch.cat(parts)
        else:
            while len(self._voxelizers) < batch:           # one workspace per in-flight frame of the batch
                self._voxelizers.append(ops.Voxelizer(self.cfg.voxel_size, self.cfg.point_cloud_range,
                                                      self.cfg.num_point_features, self.cfg.max_points_per_voxel,
                                                      self.cfg.max_voxels, device=self.device))
            outs = [self._voxelizers[b](pts, batch_idx=b, coord_cols=4, want_voxels=False, want_mean=True, sync=False)
                    for b, pts in enumerate(points_list)]
            ms = torch.cat([o[4] for o in outs]).tolist() if batch > 1 else [int(outs[0][4].item())]   # one read-back
            feats = torch.cat([o[3][:m] for o, m in zip(outs, ms)]) if batch > 1 else outs[0][3][:ms[0]]
            coords = torch.cat([o[1][:m] for o, m in zip(outs, ms)]) if batch > 1 else outs[0][1][:ms[0]]
        # A step whose activations left fp16's safe range is re-run guarded (exact), and the engine then STAYS guarded for the next
        # RANGE_STICKY_STEPS steps (each of which extends the stay while its activations are still >= 2^14): a checkpoint or scene
        # that habitually exceeds the range pays the guarded kernels' ~3 % instead of a second pass per step.
        self.range_reruns = getattr(self, "range_reruns", 0)
        self.range_guarded_steps = getattr(self, "range_guarded_steps", 0)
        self._guard_left = getattr(self, "_guard_left", 0)
        self._rb_scaled = self._guard_left > 0
        self._range_high = False
        fd, fh, fw = self._final_shape()
        self._export_pair_levels = bool(pair_levels) and proposals is not None
        while True:
            dp = (not self._rb_scaled) and self.dense_pairs_ok(batch, fh, fw)
            levels, (x, out_idx, out_shape) = self.backbone3d(feats, coords, batch, index=index0, canonical0=canonical0,
                                                              export_levels=proposals if proposals is not None else return_intermediates,
                                                              pair_rows=self.cfg.pair_rows and not self._rb_scaled, dense_pairs=dp)
            d, h, w = out_shape
            dp = dp and self.encoded_pairs
            # the BEV map: a persistent pre-zeroed buffer per batch shape -- the occupied rows (12 % at this level) are scattered in and,
            # once its one reader (the first BEV conv) is queued, zeroed again: no 36 MB-per-frame clear (round 5). Callers that keep the
            # map (return_intermediates) get a fresh tensor
            dmap = None
            if self.cfg.persistent_dense_map and not return_intermediates:
                key = ("dense_map", d, h, w, x.shape[1])
                dmap = self._bev_cache.get(key)
                if dmap is None or dmap.batch < batch:      # one map, sized for the largest batch seen; smaller batches use its leading frames
                    dmap = self._bev_cache[key] = ops.DenseMap(batch, out_shape, x.shape[1], self.device)
                dense = dmap.scatter(x, out_idx, batch).view(batch * h * w, d * x.shape[1])
            else:
                dense = ops.densify_nhwc(x, out_idx, batch, out_shape).view(batch * h * w, d * x.shape[1])
            self._dense_map = dmap
            cat, head = self.bev_and_head(dense, batch, h, w, pairs=dp)
            results = self.decode_and_nms(head, batch, h, w, raw=proposals is not None)
            if results is not None:
                break
            # an activation left fp16's safe range: the same step with every layer pre-scaling its input (exact), see _range_reset
            if self.range_reruns == 0:
                import warnings
                warnings.warn("cpd_amd: an activation reached 2^15 -- step re-run with the range-guarded f16x2 kernels; the engine "
                              "stays guarded for the next %d steps" % self.RANGE_STICKY_STEPS)
            self._rb_scaled = True
            self.range_reruns += 1
            self._guard_left = self.RANGE_STICKY_STEPS + 1
        if self._rb_scaled:
            self.range_guarded_steps += 1
            self._guard_left = self.RANGE_STICKY_STEPS if self._range_high else self._guard_left - 1
        self._rb_scaled = False
        if proposals is not None:
            return results + (levels,)
        if return_intermediates:
            if dp:                                         # exported tensors are fp32 (h + l is exact)
                x, dense, cat = ops.pairs_to_rows(x), ops.pairs_to_rows(dense), ops.pairs_to_rows(cat)
            return results, dict(voxel_features=feats, voxel_coords=coords, levels=levels,
                                 encoded=(x, out_idx, out_shape), spatial_features_nhwc=dense, bev_cat=cat,
                                 head_rows=head, dense_pairs=dp)
        return results

    __call__ = forward
