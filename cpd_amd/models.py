"""Host-side mirrors of the reference modules on the hot path, same class names, constructor
arguments, `batch_dict` keys and state_dict names, running on the HIP kernels:

    MeanVFE            cpd/models/backbones_3d/vfe/mean_vfe.py:6-61
    VoxelResBackBone8x cpd/models/backbones_3d/spconv_backbone.py:398-600 (+ SparseBasicBlock l.100-136,
                       post_act_block l.13-35)
    HeightCompression  cpd/models/backbones_2d/map_to_bev/height_compression.py:38-177 (ALIGN off)
    BaseBEVBackbone    cpd/models/backbones_2d/base_bev_backbone.py:6-122
    CenterHead         cpd/models/dense_heads/center_head.py:48-354 (forward / box generation)
    CenterPoint        cpd/models/detectors/centerpoint.py:4-50 + detector3d_template.py:22-51

These classes are the module-by-module (un-fused) drop-in path: every conv is one cpd_gather_conv
launch, BatchNorm/ReLU stay the torch modules the reference uses. `CenterPoint.to_engine()` turns the
same weights into the fused inference engine (cpd_amd/engine.py) that bench.py measures.
Registries follow the reference's `__all__[NAME]` dict convention (backbones_3d/__init__.py:3-8).
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import autograd_ops, iou3d_nms_utils, ops, train_ops
from . import spconv as _spconv_pkg
from .spconv import pytorch as spconv
from .spconv.pytorch import conv as _spc
from .spconv.pytorch.conv import default_conv_math, fold_batchnorm as _fold_batchnorm, fusable_eval as _fusable_eval


class AttrDict(dict):
    """EasyDict stand-in (cpd/config.py): attribute access + .get on nested dicts."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, AttrDict):
                self[k] = AttrDict(v)

    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def replace_feature(out, new_features):
    """cpd/utils/spconv_utils.py:58-64"""
    return out.replace_feature(new_features)


# --------------------------------------------------------------------------------------- VFE
class MeanVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, num_frames=1, **kwargs):
        super().__init__()
        self.model_cfg, self.num_point_features, self.num_frames = model_cfg, num_point_features, num_frames

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        for i in range(self.num_frames):
            fid = "" if i == 0 else str(i)
            if "voxel_features" + fid in batch_dict:       # already produced by the fused voxelizer
                continue
            voxels, num = batch_dict["voxels" + fid], batch_dict["voxel_num_points" + fid]
            mean = voxels.sum(dim=1) / torch.clamp_min(num.view(-1, 1), min=1.0).type_as(voxels)
            batch_dict["voxel_features" + fid] = mean.contiguous()
        return batch_dict


# ------------------------------------------------------------------------------- 3D backbone
def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type="subm",
                   norm_fn=None):
    if conv_type == "subm":
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == "spconv":
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    elif conv_type == "inverseconv":
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=True, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=True, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        if self.downsample is None and _fusable_eval(self) and self.bn1.track_running_stats:
            # eval, no autograd: two launches -- conv1 with bn1 + ReLU in its epilogue, conv2 with bn2 + identity + ReLU
            s1, t1 = _fold_batchnorm(self.bn1, self.conv1.bias)
            s2, t2 = _fold_batchnorm(self.bn2, self.conv2.bias)
            out = self.conv1(x, scale=s1, shift=t1, relu=True)
            # (fast eval: the identity travels as the fp16-pair rows the block was handed; conv.py decodes them if conv2 cannot take them)
            res = x._pairs if (_spc.optimistic() and x._pairs is not None) else x.features.contiguous().float()
            return self.conv2(out, scale=s2, shift=t2, residual=res, relu=True)
        identity = x
        out = self.conv1(x)
        out = replace_feature(out, self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        out = replace_feature(out, self.bn2(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = replace_feature(out, self.relu(out.features + identity.features))
        return out


def _mark_pairs_out(backbone, keep_fp32=()):
    """Fast eval (spconv.install(fast_eval=True), f16x2): every sparse conv of the backbone whose consumer is another fused sparse layer
    may leave fp16-pair rows (`pairs_out`, read by SparseConvolution.forward inside an optimistic range pass only); the layers in
    `keep_fp32` -- conv_out: HeightCompression reads fp32 rows -- never do."""
    for m in backbone.modules():
        if isinstance(m, spconv.SparseConvolution):
            m.pairs_out = not any(m is k for k in keep_fp32)


def _adopt_voxel_index(x_in, batch_dict):
    """batch_dict["voxel_index"] (optional, beyond the reference's keys): the ops.SiteIndex of `voxel_coords` over the backbone's
    sparse_shape when the voxelizer built it with the voxel list (ops.Voxelizer.batch(index_z_extra=1)) -- the level-0 index of the
    sparse tensor as it stands, instead of a second bitmap / scan / permutation pass here. Checked against the tensor it is for."""
    idx = batch_dict.get("voxel_index")
    if idx is not None and isinstance(idx, ops.SiteIndex) and idx.batch == x_in.batch_size and list(idx.shape) == list(x_in.spatial_shape):
        x_in._site_index = idx


class VoxelResBackBone8x(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, num_frames=1, **kwargs):
        super().__init__()
        self.model_cfg, self.num_frames = model_cfg, num_frames
        nf = model_cfg.NUM_FILTERS
        self.out_features = model_cfg.OUT_FEATURES
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + [1, 0, 0])]
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, nf[0], 3, padding=1, bias=False, indice_key="subm1"), norm_fn(nf[0]), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(SparseBasicBlock(nf[0], nf[0], norm_fn=norm_fn, indice_key="res1"),
                                             SparseBasicBlock(nf[0], nf[0], norm_fn=norm_fn, indice_key="res1"))
        self.conv2 = spconv.SparseSequential(
            block(nf[0], nf[1], 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2", conv_type="spconv"),
            SparseBasicBlock(nf[1], nf[1], norm_fn=norm_fn, indice_key="res2"),
            SparseBasicBlock(nf[1], nf[1], norm_fn=norm_fn, indice_key="res2"))
        self.conv3 = spconv.SparseSequential(
            block(nf[1], nf[2], 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv3", conv_type="spconv"),
            SparseBasicBlock(nf[2], nf[2], norm_fn=norm_fn, indice_key="res3"),
            SparseBasicBlock(nf[2], nf[2], norm_fn=norm_fn, indice_key="res3"))
        self.conv4 = spconv.SparseSequential(
            block(nf[2], nf[3], 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key="spconv4", conv_type="spconv"),
            SparseBasicBlock(nf[3], nf[3], norm_fn=norm_fn, indice_key="res4"),
            SparseBasicBlock(nf[3], nf[3], norm_fn=norm_fn, indice_key="res4"))
        last_pad = model_cfg.get("last_pad", 0)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(nf[3], self.out_features, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key="spconv_down2"), norm_fn(self.out_features), nn.ReLU())
        if model_cfg.get("MM", False):
            # the prototype branch's own (lighter: one residual block per level) encoder over `voxel_features1`,
            # spconv_backbone.py:456-486; run in training mode only (l.560-598)
            self.conv_input_2 = spconv.SparseSequential(
                spconv.SubMConv3d(input_channels, nf[0], 3, padding=1, bias=False, indice_key="subm1_2"), norm_fn(nf[0]), nn.ReLU())
            self.conv1_2 = spconv.SparseSequential(SparseBasicBlock(nf[0], nf[0], norm_fn=norm_fn, indice_key="res1_2"),
                                                   SparseBasicBlock(nf[0], nf[0], norm_fn=norm_fn, indice_key="res1_2"))
            self.conv2_2 = spconv.SparseSequential(
                block(nf[0], nf[1], 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv2_2", conv_type="spconv"),
                SparseBasicBlock(nf[1], nf[1], norm_fn=norm_fn, indice_key="res2_2"))
            self.conv3_2 = spconv.SparseSequential(
                block(nf[1], nf[2], 3, norm_fn=norm_fn, stride=2, padding=1, indice_key="spconv3_2", conv_type="spconv"),
                SparseBasicBlock(nf[2], nf[2], norm_fn=norm_fn, indice_key="res3_2"))
            self.conv4_2 = spconv.SparseSequential(
                block(nf[2], nf[3], 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key="spconv4_2", conv_type="spconv"),
                SparseBasicBlock(nf[3], nf[3], norm_fn=norm_fn, indice_key="res4_2"))
        self.num_point_features = self.out_features
        if model_cfg.get("RETURN_NUM_FEATURES_AS_DICT", False):
            self.num_point_features = {"x_conv1": nf[0], "x_conv2": nf[1], "x_conv3": nf[2], "x_conv4": nf[3]}
        _mark_pairs_out(self, keep_fp32=[self.conv_out[0]])

    def forward(self, batch_dict):
        x_in = spconv.SparseConvTensor(features=batch_dict["voxel_features"], indices=batch_dict["voxel_coords"].int(),
                                       spatial_shape=self.sparse_shape, batch_size=batch_dict["batch_size"])
        _adopt_voxel_index(x_in, batch_dict)
        x = self.conv_input(x_in)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        out = self.conv_out(x_conv4)
        batch_dict.update({"encoded_spconv_tensor": out, "encoded_spconv_tensor_stride": 8})
        batch_dict.update({"multi_scale_3d_features": {"x_conv1": x_conv1, "x_conv2": x_conv2, "x_conv3": x_conv3, "x_conv4": x_conv4},
                           "multi_scale_3d_strides": {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}})
        if self.training and self.model_cfg.get("MM", False):
            m_in = spconv.SparseConvTensor(features=batch_dict["voxel_features1"], indices=batch_dict["voxel_coords1"].int(),
                                           spatial_shape=self.sparse_shape, batch_size=batch_dict["batch_size"])
            m1 = self.conv1_2(self.conv_input_2(m_in))
            m2 = self.conv2_2(m1)
            m3 = self.conv3_2(m2)
            m4 = self.conv4_2(m3)
            batch_dict.update({"encoded_spconv_tensor_stride_mm": 8,
                               "multi_scale_3d_features_mm": {"x_conv1": m1, "x_conv2": m2, "x_conv3": m3, "x_conv4": m4}})
        return batch_dict


class VoxelBackBone8x(nn.Module):
    """The non-residual backbone of the reference's registry (spconv_backbone.py:138-395, backbones_3d/__init__.py:3-8): conv_input,
    one SubM block at stride 1, then per stride (2, 4, 8) a strided SparseConv3d + two SubM blocks, conv_out (3, 1, 1) / (2, 1, 1);
    same parameter names (`conv2.0.0.weight` ...), so its checkpoints load.

    forward follows the reference in both modes. Training: every stage (`transform_param` given: stage i reads `voxel_features<i>` /
    `voxel_coords<i>`) goes through the backbone on its own. Eval: the stages are laid side by side along X -- stage i's voxels
    shifted by i * sparse_shape[2] on a grid four times as wide -- so that ONE pass through the sparse convolutions serves all of
    them (l.333-359), and `decompose_tensor` (l.241-261) cuts x_conv3 / x_conv4 / out back into per-stage tensors: columns strictly
    between i * W/4 and (i + 1) * W/4 of the level's width W (the reference's open interval: column i * W/4 itself is dropped),
    shifted back; x_conv1 / x_conv2 are None in eval mode, as there."""

    def __init__(self, model_cfg, input_channels, grid_size, num_frames=1, **kwargs):
        super().__init__()
        self.model_cfg, self.num_frames = model_cfg, num_frames
        nf = model_cfg.NUM_FILTERS
        self.out_features = model_cfg.OUT_FEATURES
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + [1, 0, 0])]
        block = post_act_block

        def encoder(sfx):
            conv_input = spconv.SparseSequential(
                spconv.SubMConv3d(input_channels, nf[0], 3, padding=1, bias=False, indice_key="subm1" + sfx), norm_fn(nf[0]), nn.ReLU())
            conv1 = spconv.SparseSequential(block(nf[0], nf[0], 3, norm_fn=norm_fn, padding=1, indice_key="subm1" + sfx))
            stages = []
            for lvl, pad in ((1, 1), (2, 1), (3, (0, 1, 1))):
                key = "%d%s" % (lvl + 1, sfx)
                stages.append(spconv.SparseSequential(
                    block(nf[lvl - 1], nf[lvl], 3, norm_fn=norm_fn, stride=2, padding=pad, indice_key="spconv" + key, conv_type="spconv"),
                    block(nf[lvl], nf[lvl], 3, norm_fn=norm_fn, padding=1, indice_key="subm" + key),
                    block(nf[lvl], nf[lvl], 3, norm_fn=norm_fn, padding=1, indice_key="subm" + key)))
            return [conv_input, conv1] + stages

        self.conv_input, self.conv1, self.conv2, self.conv3, self.conv4 = encoder("")
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(nf[3], self.out_features, (3, 1, 1), stride=(2, 1, 1), padding=model_cfg.get("last_pad", 0), bias=False,
                                indice_key="spconv_down2"), norm_fn(self.out_features), nn.ReLU())
        if model_cfg.get("MM", False):           # declared like the reference (l.194-227); its forward never runs them either
            self.conv_input_2, self.conv1_2, self.conv2_2, self.conv3_2, self.conv4_2 = encoder("_2")
        self.num_point_features = self.out_features
        if model_cfg.get("RETURN_NUM_FEATURES_AS_DICT", False):
            self.num_point_features = {"x_conv1": nf[0], "x_conv2": nf[1], "x_conv3": nf[2], "x_conv4": nf[3]}

    def _encode(self, feats, coords, shape, batch_size):
        x = self.conv_input(spconv.SparseConvTensor(features=feats, indices=coords.int(), spatial_shape=shape, batch_size=batch_size))
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        return x1, x2, x3, x4, self.conv_out(x4)

    @staticmethod
    def decompose_tensor(tensor, i, batch_size):
        """stage i's share of a side-by-side tensor (reference l.241-261)"""
        quarter = tensor.spatial_shape[2] // 4
        xs = tensor.indices[:, 3]
        keep = (xs > i * quarter) & (xs < (i + 1) * quarter)
        idx = tensor.indices[keep].clone()
        idx[:, 3] -= i * quarter
        return spconv.SparseConvTensor(features=tensor.features[keep], indices=idx.int(),
                                       spatial_shape=[tensor.spatial_shape[0], tensor.spatial_shape[1], quarter], batch_size=batch_size)

    def forward(self, batch_dict):
        stages = batch_dict["transform_param"].shape[1] if "transform_param" in batch_dict else 1
        batch_size = batch_dict["batch_size"]
        sid = lambda i: "" if i == 0 else str(i)
        strides = {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}
        if self.training:
            for i in range(stages):
                x1, x2, x3, x4, out = self._encode(batch_dict["voxel_features" + sid(i)], batch_dict["voxel_coords" + sid(i)],
                                                   self.sparse_shape, batch_size)
                batch_dict.update({"encoded_spconv_tensor" + sid(i): out, "encoded_spconv_tensor_stride" + sid(i): 8,
                                   "multi_scale_3d_features" + sid(i): {"x_conv1": x1, "x_conv2": x2, "x_conv3": x3, "x_conv4": x4},
                                   "multi_scale_3d_strides" + sid(i): dict(strides)})
            return batch_dict
        feats, coords = [], []
        for i in range(stages):
            feats.append(batch_dict["voxel_features" + sid(i)])
            c = batch_dict["voxel_coords" + sid(i)].clone()
            c[:, 3] += i * self.sparse_shape[2]
            coords.append(c)
        wide = [self.sparse_shape[0], self.sparse_shape[1], self.sparse_shape[2] * 4]
        _, _, x3, x4, out = self._encode(torch.cat(feats, 0), torch.cat(coords), wide, batch_size)
        for i in range(stages):
            batch_dict.update({"encoded_spconv_tensor" + sid(i): self.decompose_tensor(out, i, batch_size),
                               "encoded_spconv_tensor_stride" + sid(i): 8,
                               "multi_scale_3d_features" + sid(i): {"x_conv1": None, "x_conv2": None,
                                                                    "x_conv3": self.decompose_tensor(x3, i, batch_size),
                                                                    "x_conv4": self.decompose_tensor(x4, i, batch_size)},
                               "multi_scale_3d_strides" + sid(i): dict(strides)})
        return batch_dict


# ------------------------------------------------------------------------------------- BEV
class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg.NUM_BEV_FEATURES

    def forward(self, batch_dict):
        sp = batch_dict["encoded_spconv_tensor"]
        if _fusable_eval(self) and sp.features.is_cuda:
            # same (N, C*D, H, W) tensor, channel = c*D + z, in channels_last memory: the BEV convs read it without a transpose pass
            rows = ops.densify_nhwc_cd(sp.features.float(), sp.indices, sp.batch_size, sp.spatial_shape)
            batch_dict["spatial_features"] = _tag(rows.permute(0, 3, 1, 2), ops.tagged_range(sp.features))
            batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
            return batch_dict
        dense = sp.dense()
        n, c, d, h, w = dense.shape
        batch_dict["spatial_features"] = dense.view(n, c * d, h, w)
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict


def _rows(x):
    """(B,C,H,W) -> channels-last rows [B*H*W, C] (no copy when x already is channels_last)."""
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).contiguous().view(b * h * w, c)


def _nchw(rows, b, h, w):
    return rows.view(b, h, w, rows.shape[1]).permute(0, 3, 1, 2)   # NCHW view, channels_last strides


def _range_in_out(x, c_in, math, out_rb=None):
    """f16x2 range guard of the fused module path: (block of the input map -- the attribute its producer left, else measured --,
    block this launch fills: the caller's `out_rb` or a fresh one); (None, None) for the other arithmetics."""
    if math != "f16x2":
        return None, None
    if _spc.optimistic():                       # fast eval: unscaled kernels, max |out| recorded in the pass's pool (spconv/pytorch/conv.py)
        return None, _spc.record_block()
    rb_in = None
    if c_in % 32 == 0:
        rb_in = ops.tagged_range(x)
        if rb_in is None:
            rb_in = ops.absmax_rows(_rows(x.float()), c_in)
    return rb_in, (out_rb if out_rb is not None else ops.absmax_blocks(1, x.device)[0])


def _tag(t, rb):
    return ops.tag_range(t, rb)


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters (state_dict compatible), forward on cpd_gather_conv with a dense pixel
    rulebook. k x k, one stride, zero padding."""
    _tables = {}

    @property
    def math(self):
        """this layer's arithmetic: its own `conv_math` attribute, else the package default (spconv.install(conv_math=...))"""
        m = getattr(self, "conv_math", None)
        return m if m is not None else default_conv_math()

    def _table(self, b, h, w, p, device):
        k, s = self.kernel_size[0], self.stride[0]
        key = (b, h, w, k, s, p, device)
        if key not in Conv2d._tables:
            Conv2d._tables[key] = ops.rulebook_conv2d(b, h, w, k, k, s, p, device)
        return key, Conv2d._tables[key]

    def _packed(self):
        k = self.kernel_size[0]
        ver = (self.weight._version, self.weight.data_ptr())
        if getattr(self, "_pk_ver", None) != ver:
            self._pk = ops.pack_weight(self.weight.detach().permute(2, 3, 1, 0).reshape(k * k, self.in_channels, self.out_channels).contiguous())
            self._pk_ver = ver
        return self._pk

    def forward_fused(self, x, scale=None, shift=None, relu=False, extra_pad=0, out=None, out_rb=None):
        """Inference launch with the epilogue fused: zero padding of a preceding ZeroPad2d folded into the pixel table
        (extra_pad), folded BatchNorm2d (scale, shift; shift already carries the bias) and ReLU in the conv kernel. `x`:
        (B, C, H, W) in any memory format (channels_last = no copy); `out`: optional [B*Ho*Wo, >= Cout] row view to write."""
        b, c, h, w = x.shape
        k = self.kernel_size[0]
        _, (nbr, ho, wo) = self._table(b, h, w, self.padding[0] + extra_pad, x.device)
        shift = ops.epilogue_shift(scale, shift, self.bias)
        rb_in, rb_out = _range_in_out(x, c, self.math, out_rb)
        rows = ops.gather_conv(_rows(x.float()), c, self._packed(), nbr, k * k, b * ho * wo, self.out_channels, scale, shift, None, relu,
                               out=out, dense=True, math=self.math, in_absmax=rb_in, out_absmax=rb_out)
        return _tag(_nchw(rows[:, :self.out_channels] if out is not None else rows, b, ho, wo), rb_out)

    def forward(self, x):
        assert x.is_cuda, "cpd_amd.models.Conv2d runs on the GPU only"
        b, c, h, w = x.shape
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        key, (nbr, ho, wo) = self._table(b, h, w, p, x.device)
        self._packed()

        def adjoint():          # transposed pixel table of a strided / unpadded conv (input gradient)
            tkey = key + ("t",)
            if tkey not in Conv2d._tables:
                Conv2d._tables[tkey] = train_ops.rulebook_conv2d_transpose(b, h, w, k, k, s, p, x.device)
            return Conv2d._tables[tkey]

        same = s == 1 and 2 * p == k - 1
        spec = autograd_ops.ConvSpec(nbr, k * k, b * ho * wo, dense=True, math=self.math,
                                     mode="same" if same else "strided", adjoint=adjoint, packed=self._pk)
        w_kio = self.weight.permute(2, 3, 1, 0).reshape(k * k, c, self.out_channels)
        out = autograd_ops.gather_conv(_rows(x.float()), w_kio, self.bias, spec)
        return _nchw(out, b, ho, wo)


class ConvTranspose2d(nn.ConvTranspose2d):
    """kernel == stride, no padding (base_bev_backbone.py:52-56): one 1x1 GEMM over the k*k taps."""
    _maps = {}
    math = Conv2d.math

    def _packed(self):
        u = self.kernel_size[0]
        ver = (self.weight._version, self.weight.data_ptr())
        if getattr(self, "_pk_ver", None) != ver:
            self._pk = ops.pack_weight(self.weight.detach().permute(0, 2, 3, 1).reshape(1, self.in_channels, u * u * self.out_channels).contiguous())
            self._pk_ver = ver
        return self._pk

    @staticmethod
    def _up_map(b, h, w, u, device):
        key = (b, h, w, u, device)
        if key not in ConvTranspose2d._maps:
            H, W = h * u, w * u
            bi = torch.arange(b, device=device).view(-1, 1, 1)
            yy = torch.arange(h, device=device).view(1, -1, 1)
            xx = torch.arange(w, device=device).view(1, 1, -1)
            maps = [((bi * H + u * yy + a) * W + u * xx + bb).reshape(-1) for a in range(u) for bb in range(u)]
            ConvTranspose2d._maps[key] = torch.stack(maps).to(torch.int32).contiguous()
        return ConvTranspose2d._maps[key]

    def forward_fused(self, x, scale=None, shift=None, relu=False, extra_pad=0, out=None, out_rb=None):
        """Inference launch: the u*u taps as column groups of one 1x1 GEMM scattered to the upsampled rows, folded BatchNorm2d
        + ReLU in the epilogue; `out`: optional [B*H*W, >= Cout] row view (e.g. a column block of the concat buffer)."""
        assert extra_pad == 0
        b, c, h, w = x.shape
        u = self.kernel_size[0]
        assert self.stride[0] == u and self.padding[0] == 0
        H, W, co = h * u, w * u, self.out_channels
        shift = ops.epilogue_shift(scale, shift, self.bias)
        if out is None:
            out = torch.empty((b * H * W, co), dtype=torch.float32, device=x.device)
        rep = (lambda t: t.repeat(u * u) if t is not None and u > 1 else t)
        rb_in, rb_out = _range_in_out(x, c, self.math, out_rb)
        if u == 1:
            ops.gather_conv(_rows(x.float()), c, self._packed(), None, 1, b * h * w, co, scale, shift, None, relu, out=out, dense=True,
                            math=self.math, in_absmax=rb_in, out_absmax=rb_out)
        else:
            ops.gather_conv(_rows(x.float()), c, self._packed(), None, 1, b * h * w, u * u * co, rep(scale), rep(shift), None, relu,
                            out=out, out_row_map=self._up_map(b, h, w, u, x.device), out_col_group=co, dense=True, math=self.math,
                            in_absmax=rb_in, out_absmax=rb_out)
        return _tag(_nchw(out[:, :co], b, H, W), rb_out)

    def forward(self, x):
        b, c, h, w = x.shape
        u = self.kernel_size[0]
        assert self.stride[0] == u and self.padding[0] == 0
        self._packed()
        H, W = h * u, w * u
        w_kio = self.weight.permute(0, 2, 3, 1).reshape(1, c, u * u * self.out_channels)
        math = self.math
        if u == 1:
            spec = autograd_ops.ConvSpec(None, 1, b * h * w, dense=True, math=math, mode="same", packed=self._pk)
            out = autograd_ops.gather_conv(_rows(x.float()), w_kio, self.bias, spec)
        else:
            spec = autograd_ops.ConvSpec(None, 1, b * h * w, dense=True, math=math, mode="up", up_map=self._up_map(b, h, w, u, x.device),
                                         up=u, n_up=b * H * W, packed=self._pk)
            out = autograd_ops.gather_conv(_rows(x.float()), w_kio, self.bias, spec)
        return _nchw(out, b, H, W)


class DenseSequential(nn.Sequential):
    """nn.Sequential (same children, same state_dict names) whose eval / no-autograd forward fuses what the reference writes as
    separate modules: ZeroPad2d(p) -> Conv2d becomes the conv's own padding, Conv2d | ConvTranspose2d -> BatchNorm2d [-> ReLU]
    one launch with the folded statistics and the ReLU in the conv epilogue (base_bev_backbone.py:31-59, center_head.py:21-27,
    73-80). Maps stay channels-last in memory between layers. In training mode, or under autograd, it is nn.Sequential."""

    def forward(self, x, out=None, out_rb=None):
        mods = list(self)
        if not _fusable_eval(*mods):
            y = super().forward(x)
            if out is not None:                       # the caller's buffer is written on EVERY path
                out[:, :y.shape[1]].copy_(_rows(y.float()))
            return y
        wrote = False
        i, pad = 0, 0
        while i < len(mods):
            m = mods[i]
            i += 1
            if isinstance(m, nn.ZeroPad2d) and i < len(mods) and isinstance(mods[i], Conv2d) and len(set(m.padding)) == 1:
                pad = int(m.padding[0])
                continue
            if isinstance(m, (Conv2d, ConvTranspose2d)):
                scale = shift = None
                relu = False
                if i < len(mods) and isinstance(mods[i], nn.BatchNorm2d) and mods[i].track_running_stats:
                    scale, shift = _fold_batchnorm(mods[i], m.bias)
                    i += 1
                    if i < len(mods) and isinstance(mods[i], nn.ReLU):
                        relu = True
                        i += 1
                last = i == len(mods)
                x = m.forward_fused(x, scale, shift, relu, extra_pad=pad, out=out if last else None, out_rb=out_rb if last else None)
                wrote = wrote or last
                pad = 0
                continue
            x = m(x)
        if out is not None and not wrote:             # the sequence did not end in a fused conv: copy what it produced
            out[:, :x.shape[1]].copy_(_rows(x.float()))
        return x


class BaseBEVBackbone(nn.Module):
    def __init__(self, model_cfg, num_frames=1, input_channels=256, **kwargs):
        super().__init__()
        self.model_cfg, self.num_frames = model_cfg, num_frames
        layer_nums, layer_strides, num_filters = model_cfg.LAYER_NUMS, model_cfg.LAYER_STRIDES, model_cfg.NUM_FILTERS
        num_up, up_strides = model_cfg.NUM_UPSAMPLE_FILTERS, model_cfg.UPSAMPLE_STRIDES
        c_in_list = [input_channels, *num_filters[:-1]]
        self.blocks, self.deblocks = nn.ModuleList(), nn.ModuleList()
        for idx in range(len(layer_nums)):
            layers = [nn.ZeroPad2d(1), Conv2d(c_in_list[idx], num_filters[idx], kernel_size=3, stride=layer_strides[idx], padding=0, bias=False),
                      nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()]
            for _ in range(layer_nums[idx]):
                layers.extend([Conv2d(num_filters[idx], num_filters[idx], kernel_size=3, padding=1, bias=False),
                               nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()])
            self.blocks.append(DenseSequential(*layers))
            self.deblocks.append(DenseSequential(
                ConvTranspose2d(num_filters[idx], num_up[idx], up_strides[idx], stride=up_strides[idx], bias=False),
                nn.BatchNorm2d(num_up[idx], eps=1e-3, momentum=0.01), nn.ReLU()))
        self.num_bev_features_post = sum(num_up)

    def forward(self, data_dict):
        x = data_dict["spatial_features"]
        if len(self.blocks) > 1 and _fusable_eval(self):
            # eval, no autograd: every deblock writes its column block of ONE channels-last concat buffer (no torch.cat pass)
            cat, col = None, 0
            cat_rb = ops.absmax_blocks(1, x.device)[0] if self.deblocks[0][0].math == "f16x2" else None    # one block for the concat
            for i in range(len(self.blocks)):
                x = self.blocks[i](x)
                de = self.deblocks[i]
                c_up = de[0].out_channels
                if cat is None:
                    b, _, h, w = x.shape
                    u = de[0].kernel_size[0]
                    H, W = h * u, w * u
                    cat = torch.empty((b * H * W, self.num_bev_features_post), dtype=torch.float32, device=x.device)
                de(x, out=cat[:, col:col + c_up], out_rb=cat_rb)
                col += c_up
            data_dict["st_features_2d"] = _tag(_nchw(cat, b, H, W), cat_rb)
            return data_dict
        ups = []
        for i in range(len(self.blocks)):
            x = self.blocks[i](x)
            ups.append(self.deblocks[i](x))
        data_dict["st_features_2d"] = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        return data_dict


# -------------------------------------------------------------------------------- CenterHead
class SeparateHead(nn.Module):
    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for cur_name in sep_head_dict:
            out_c, num_conv = sep_head_dict[cur_name]["out_channels"], sep_head_dict[cur_name]["num_conv"]
            fc = []
            for _ in range(num_conv - 1):
                fc.append(DenseSequential(Conv2d(input_channels, input_channels, kernel_size=3, stride=1, padding=1, bias=use_bias),
                                          nn.BatchNorm2d(input_channels), nn.ReLU()))
            fc.append(Conv2d(input_channels, out_c, kernel_size=3, stride=1, padding=1, bias=True))
            fc = DenseSequential(*fc)
            if "hm" in cur_name:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.kaiming_normal_(m.weight.data)
                        if m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            self.__setattr__(cur_name, fc)

    def _fused_images(self, dev):
        """All branches' first convs side by side (C -> n_heads * C, BatchNorm folded) and their output convs as one
        block-diagonal conv (n_heads * C -> sum of the heads' channels): two launches for the whole SeparateHead, as in the
        engine. Rebuilt when a parameter or a BatchNorm statistic changes."""
        tensors = list(self.parameters()) + list(self.buffers())
        key = tuple((t._version, t.data_ptr()) for t in tensors)
        cached = getattr(self, "_fused", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        names = list(self.sep_head_dict)
        w1, s1, t1, slices = [], [], [], {}
        c = self.__getattr__(names[0])[0][0].in_channels
        n_out = sum(self.__getattr__(n)[1].out_channels for n in names)
        w2 = torch.zeros(9, c * len(names), n_out)
        b2 = torch.zeros(n_out)
        col = 0
        with torch.no_grad():
            for hi, name in enumerate(names):
                fc = self.__getattr__(name)
                conv, bn, last = fc[0][0], fc[0][1], fc[1]
                w1.append(conv.weight.detach().float().cpu().permute(2, 3, 1, 0).reshape(9, c, c))
                sc, sh = _fold_batchnorm(bn, conv.bias)
                s1.append(sc.cpu()); t1.append(sh.cpu())
                co = last.out_channels
                w2[:, hi * c:(hi + 1) * c, col:col + co] = last.weight.detach().float().cpu().permute(2, 3, 1, 0).reshape(9, c, co)
                b2[col:col + co] = last.bias.detach().float().cpu()
                slices[name] = (col, co)
                col += co
            img = dict(w1=ops.pack_weight(torch.cat(w1, dim=2).to(dev).contiguous()), s1=torch.cat(s1).to(dev), t1=torch.cat(t1).to(dev),
                       w2=ops.pack_weight(w2.to(dev).contiguous()), b2=b2.to(dev), c=c, c1=c * len(names), n_out=n_out, slices=slices,
                       ld=16 * ((n_out + 15) // 16))
        self._fused = (key, img)
        return img

    def _fusable(self):
        if not _fusable_eval(self):
            return False
        for name in self.sep_head_dict:
            fc = self.__getattr__(name)
            if len(fc) != 2 or not isinstance(fc[0], DenseSequential) or not fc[0][1].track_running_stats:
                return False
        return True

    def forward(self, x):
        if self._fusable():
            b, c, h, w = x.shape
            img = self._fused_images(x.device)
            nbr, _, _ = ops.rulebook_conv2d_cached(b, h, w, x.device)
            n = b * h * w
            math = self.__getattr__(next(iter(self.sep_head_dict)))[1].math
            rb_in, rb_h1 = _range_in_out(x, c, math)
            h1 = ops.gather_conv(_rows(x.float()), c, img["w1"], nbr, 9, n, img["c1"], img["s1"], img["t1"], None, True, dense=True, math=math,
                                 in_absmax=rb_in, out_absmax=rb_h1)
            rows = torch.empty((n, img["ld"]), dtype=torch.float32, device=x.device)
            ops.gather_conv(h1, img["c1"], img["w2"], nbr, 9, n, img["n_out"], None, img["b2"], None, False, out=rows, dense=True, math=math,
                            in_absmax=None if _spc.optimistic() else rb_h1)
            maps = HeadMaps((name, _nchw(rows[:, c0:c0 + co], b, h, w)) for name, (c0, co) in img["slices"].items())
            maps.rows, maps.slices, maps.ld = rows, img["slices"], img["ld"]
            return maps
        return {name: self.__getattr__(name)(x) for name in self.sep_head_dict}


class HeadMaps(dict):
    """The maps of one SeparateHead as NCHW views of ONE channels-last row block (`rows` [B*H*W, ld], `slices` name -> (first
    column, channels)): what the fused eval forward returns, and what lets decode + NMS run batched on the block."""


class CenterHead(nn.Module):
    def __init__(self, model_cfg, num_frames, input_channels, num_class, class_names, grid_size, point_cloud_range,
                 voxel_size, predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg, self.num_class, self.grid_size = model_cfg, num_class, grid_size
        self.point_cloud_range, self.voxel_size = point_cloud_range, voxel_size
        self.feature_map_stride = model_cfg.TARGET_ASSIGNER_CONFIG.get("FEATURE_MAP_STRIDE", None)
        self.class_names = class_names
        self.class_names_each_head, self.class_id_mapping_each_head = [], []
        for cur in model_cfg.CLASS_NAMES_EACH_HEAD:
            self.class_names_each_head.append([x for x in cur if x in class_names])
            self.class_id_mapping_each_head.append(torch.tensor([class_names.index(x) for x in cur if x in class_names]))
        use_bias = model_cfg.get("USE_BIAS_BEFORE_NORM", False)
        sc = model_cfg.SHARED_CONV_CHANNEL
        self.shared_conv = DenseSequential(Conv2d(input_channels, sc, 3, stride=1, padding=1, bias=use_bias), nn.BatchNorm2d(sc), nn.ReLU())
        self.heads_list = nn.ModuleList()
        for cur in self.class_names_each_head:
            head_dict = {k: dict(v) for k, v in model_cfg.SEPARATE_HEAD_CFG.HEAD_DICT.items()}
            head_dict["hm"] = dict(out_channels=len(cur), num_conv=model_cfg.NUM_HM_CONV)
            self.heads_list.append(SeparateHead(sc, head_dict, init_bias=-2.19, use_bias=use_bias))
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}

    def generate_predicted_boxes(self, batch_size, pred_dicts):
        """center_head.py:252-303 with decode + NMS on the device (cpd_center_decode, cpd_nms_rotated)."""
        pp = self.model_cfg.POST_PROCESSING
        if len(pred_dicts) == 1 and isinstance(pred_dicts[0], HeadMaps) and pp.NMS_CONFIG.NMS_TYPE == "nms_gpu" \
                and pp.MAX_OBJ_PER_SAMPLE <= pp.NMS_CONFIG.NMS_PRE_MAXSIZE:
            return self._predicted_boxes_batched(batch_size, pred_dicts[0])
        ret = [{"pred_boxes": [], "pred_scores": [], "pred_labels": []} for _ in range(batch_size)]
        for idx, pd in enumerate(pred_dicts):
            hm = pd["hm"]
            nc, h, w = hm.shape[1:]
            maps = {k: _rows(pd[k].float()) for k in ("hm", "center", "center_z", "dim", "rot")}   # [B*H*W, c]
            mapping = self.class_id_mapping_each_head[idx].to(hm.device)
            for b in range(batch_size):
                sl = slice(b * h * w, (b + 1) * h * w)
                views = [maps[k][sl] for k in ("hm", "center", "center_z", "dim", "rot")]
                # each map is its own [HW, c] row tensor: pixel stride = its channel count
                boxes, scores, labels, n = _decode_separate(views, nc, h, w, pp, self.feature_map_stride,
                                                            self.voxel_size, self.point_cloud_range)
                labels = mapping[labels.long()]
                if pp.NMS_CONFIG.NMS_TYPE != "circle_nms" and n > 0:
                    s_top, ind = torch.topk(scores, k=min(pp.NMS_CONFIG.NMS_PRE_MAXSIZE, n))
                    keep, _ = getattr(iou3d_nms_utils, pp.NMS_CONFIG.NMS_TYPE)(boxes[ind][:, 0:7], s_top, pp.NMS_CONFIG.NMS_THRESH)
                    sel = ind[keep[:pp.NMS_CONFIG.NMS_POST_MAXSIZE]]
                    boxes, scores, labels = boxes[sel], scores[sel], labels[sel]
                ret[b]["pred_boxes"].append(boxes); ret[b]["pred_scores"].append(scores); ret[b]["pred_labels"].append(labels)
        for b in range(batch_size):
            ret[b]["pred_boxes"] = torch.cat(ret[b]["pred_boxes"], dim=0)
            ret[b]["pred_scores"] = torch.cat(ret[b]["pred_scores"], dim=0)
            ret[b]["pred_labels"] = torch.cat(ret[b]["pred_labels"], dim=0) + 1
        return ret

    def _predicted_boxes_batched(self, batch_size, maps):
        """The same decode + class-agnostic NMS for ALL samples in five launches on the head's row block, one host read (the
        per-sample counts): scores leave the decode sorted, so nms_gpu's topk(NMS_PRE_MAXSIZE) + sort are identities while
        MAX_OBJ_PER_SAMPLE <= NMS_PRE_MAXSIZE (checked by the caller)."""
        pp = self.model_cfg.POST_PROCESSING
        rows, sl, ld = maps.rows, maps.slices, maps.ld
        nc, h, w = maps["hm"].shape[1:]
        boxes, scores, labels, counts = ops.center_decode(
            rows[:, sl["hm"][0]:], rows[:, sl["center"][0]:], rows[:, sl["center_z"][0]:], rows[:, sl["dim"][0]:], rows[:, sl["rot"][0]:],
            ld, 1, nc, h, w, pp.MAX_OBJ_PER_SAMPLE, float(self.feature_map_stride), list(self.voxel_size)[:2],
            list(self.point_cloud_range)[:2], list(pp.POST_CENTER_LIMIT_RANGE), pp.SCORE_THRESH, sync=False, batch=batch_size,
            sample_stride=h * w * ld)
        mapping = self.class_id_mapping_each_head[0]
        assert mapping.tolist() == list(range(len(mapping))), "the batched path keeps head-local class ids (one head over all classes)"
        keep, num_keep = ops.nms_batch(boxes, counts, pp.NMS_CONFIG.NMS_THRESH)
        ob, os_, ol, on = ops.select_boxes(boxes, scores, labels, keep, num_keep, pp.NMS_CONFIG.NMS_POST_MAXSIZE, label_offset=1)
        ns = on.tolist()
        return [{"pred_boxes": ob[b, :ns[b]], "pred_scores": os_[b, :ns[b]], "pred_labels": ol[b, :ns[b]]} for b in range(batch_size)]

    def assign_targets(self, gt_boxes, feature_map_size=None, **kwargs):
        """center_head.py:159-219 for every head, vectorised on the device (cpd_amd.center_loss;
        pinned on goldens of the reference's assign_target_of_single_head). gt_boxes (B, M, 8)."""
        from . import center_loss
        tc = self.model_cfg.TARGET_ASSIGNER_CONFIG
        ret = {"heatmaps": [], "target_boxes": [], "inds": [], "masks": []}
        names = ["bg"] + list(self.class_names)
        for cur in self.class_names_each_head:
            # keep this head's boxes, relabelled 1..len(cur) (center_head.py:180-196)
            cls = gt_boxes[..., -1].long()
            new = torch.zeros_like(cls)
            for j, cn in enumerate(cur):
                new = torch.where(cls == names.index(cn), torch.full_like(cls, j + 1), new)
            g = torch.cat([gt_boxes[..., :-1], new.unsqueeze(-1).to(gt_boxes.dtype)], dim=-1)
            g = g * (new > 0).unsqueeze(-1).to(g.dtype)
            heat, tgt, inds, masks = center_loss.assign_targets(
                g, tuple(feature_map_size), self.point_cloud_range, self.voxel_size, len(cur), tc.FEATURE_MAP_STRIDE,
                num_max_objs=tc.NUM_MAX_OBJS, gaussian_overlap=tc.GAUSSIAN_OVERLAP, min_radius=tc.MIN_RADIUS)
            ret["heatmaps"].append(heat); ret["target_boxes"].append(tgt); ret["inds"].append(inds); ret["masks"].append(masks)
        return ret

    def get_loss(self):
        """center_head.py:225-250 (differentiable: the head maps carry autograd history through the C-ABI convs; the fused,
        graph-free train step with hand-written backward is cpd_amd.train_engine, CenterPoint.to_trainer())."""
        from . import center_loss
        pds, td = self.forward_ret_dict["pred_dicts"], self.forward_ret_dict["target_dicts"]
        lw = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        order = list(self.model_cfg.SEPARATE_HEAD_CFG.HEAD_ORDER)
        tb_dict, loss = {}, 0
        for idx, pd in enumerate(pds):
            b, nc, h, w = pd["hm"].shape
            rows = torch.cat([pd[k] for k in order] + [pd["hm"]], dim=1).permute(0, 2, 3, 1).reshape(b * h * w, -1)
            hm_col = rows.shape[1] - nc
            l, parts = center_loss.center_head_loss(rows, b, h, w, td["heatmaps"][idx], td["target_boxes"][idx], td["inds"][idx],
                                                    td["masks"][idx], nc, hm_col=hm_col, code_weights=lw["code_weights"],
                                                    loc_weight=lw["loc_weight"], cls_weight=lw["cls_weight"])
            loss = loss + l
            tb_dict["hm_loss_head_%d" % idx] = parts["hm_loss"].item()
            tb_dict["loc_loss_head_%d" % idx] = parts["loc_loss"].item()
        tb_dict["rpn_loss"] = float(loss)
        return loss, tb_dict

    @staticmethod
    def reorder_rois_for_refining(batch_size, pred_dicts):
        n_max = max(1, max(len(d["pred_boxes"]) for d in pred_dicts))
        pb = pred_dicts[0]["pred_boxes"]
        rois = pb.new_zeros((batch_size, n_max, pb.shape[-1]))
        roi_scores = pb.new_zeros((batch_size, n_max))
        roi_labels = pb.new_zeros((batch_size, n_max)).long()
        for b in range(batch_size):
            n = len(pred_dicts[b]["pred_boxes"])
            rois[b, :n] = pred_dicts[b]["pred_boxes"]; roi_scores[b, :n] = pred_dicts[b]["pred_scores"]
            roi_labels[b, :n] = pred_dicts[b]["pred_labels"]
        return rois, roi_scores, roi_labels

    def forward(self, data_dict):
        x = self.shared_conv(data_dict["st_features_2d"])
        pred_dicts = [head(x) for head in self.heads_list]
        self.forward_ret_dict["pred_dicts"] = pred_dicts
        if self.training:
            h, w = x.shape[2], x.shape[3]
            self.forward_ret_dict["target_dicts"] = self.assign_targets(data_dict["gt_boxes"], feature_map_size=(h, w))
            if not self.predict_boxes_when_training:
                return data_dict
        boxes = self.generate_predicted_boxes(data_dict["batch_size"], pred_dicts)
        if self.predict_boxes_when_training:
            rois, roi_scores, roi_labels = self.reorder_rois_for_refining(data_dict["batch_size"], boxes)
            data_dict.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels, has_class_labels=True)
        else:
            data_dict["final_box_dicts"] = boxes
        return data_dict


def _decode_separate(views, nc, h, w, pp, stride, voxel_size, pcr):
    hm, center, cz, dim, rot = views
    # the five maps live in separate tensors; cpd_center_decode takes one stride pair, so gather them
    # into one channels-last row block (11 floats per pixel -- 1.5 MB at 188x188)
    rows = torch.cat([center, cz, dim, rot, hm], dim=1).contiguous()
    ld = rows.shape[1]
    return ops.center_decode(rows[:, 8:], rows[:, 0:], rows[:, 2:], rows[:, 3:], rows[:, 6:], ld, 1, nc, h, w,
                             pp.MAX_OBJ_PER_SAMPLE, float(stride), list(voxel_size)[:2], list(pcr)[:2],
                             list(pp.POST_CENTER_LIMIT_RANGE), pp.SCORE_THRESH)


# --------------------------------------------------------------------------------- detector
def _anchor_heads():
    from . import anchor_head                        # (imports this module's Conv2d lazily)
    from . import roi_head_train, roi_pool
    return {"AnchorHeadSingle": anchor_head.AnchorHeadSingle, "AnchorHeadSingleV2": anchor_head.AnchorHeadSingleV2,
            "VoxelRCNNHead": roi_pool.VoxelRCNNHead, "VoxelRCNNProtoHead": roi_head_train.VoxelRCNNProtoHead}


class _Registry(dict):
    """`__all__[NAME]` like the reference's registries (dense_heads/__init__.py, roi_heads/__init__.py); the anchor heads and
    the RoI heads resolve on first use."""

    def __missing__(self, name):
        heads = _anchor_heads()
        if name in heads:
            self.update(heads)
            return heads[name]
        raise KeyError(name)


__all__ = _Registry({"MeanVFE": MeanVFE, "VoxelResBackBone8x": VoxelResBackBone8x, "VoxelBackBone8x": VoxelBackBone8x,
                     "HeightCompression": HeightCompression,
                     "BaseBEVBackbone": BaseBEVBackbone, "CenterHead": CenterHead})


def waymo_centerpoint_cfg():
    """MODEL section of tools/cfgs/models/waymo_unsupervised/voxel_rcnn_cproto_center.yaml:13-80 (one-stage part)."""
    return AttrDict(
        NAME="CenterPoint",
        VFE=dict(NAME="MeanVFE"),
        BACKBONE_3D=dict(NAME="VoxelResBackBone8x", NUM_FILTERS=[16, 32, 64, 128], RETURN_NUM_FEATURES_AS_DICT=True, OUT_FEATURES=128),
        MAP_TO_BEV=dict(NAME="HeightCompression", NUM_BEV_FEATURES=256),
        BACKBONE_2D=dict(NAME="BaseBEVBackbone", LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2], NUM_FILTERS=[128, 256],
                         UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[256, 256]),
        DENSE_HEAD=dict(NAME="CenterHead", CLASS_AGNOSTIC=False, CLASS_NAMES_EACH_HEAD=[["Vehicle", "Pedestrian", "Cyclist"]],
                        SHARED_CONV_CHANNEL=64, USE_BIAS_BEFORE_NORM=True, NUM_HM_CONV=2,
                        SEPARATE_HEAD_CFG=dict(HEAD_ORDER=["center", "center_z", "dim", "rot"],
                                               HEAD_DICT={"center": dict(out_channels=2, num_conv=2), "center_z": dict(out_channels=1, num_conv=2),
                                                          "dim": dict(out_channels=3, num_conv=2), "rot": dict(out_channels=2, num_conv=2)}),
                        TARGET_ASSIGNER_CONFIG=dict(FEATURE_MAP_STRIDE=8, NUM_MAX_OBJS=500, GAUSSIAN_OVERLAP=0.1, MIN_RADIUS=2),
                        LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, code_weights=[1.0] * 8)),
                        POST_PROCESSING=dict(SCORE_THRESH=0.1, POST_CENTER_LIMIT_RANGE=[-75.2, -75.2, -2, 75.2, 75.2, 4],
                                             MAX_OBJ_PER_SAMPLE=500,
                                             NMS_CONFIG=dict(NMS_TYPE="nms_gpu", NMS_THRESH=0.8, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500))))


class CenterPoint(nn.Module):
    """Detector3DTemplate.build_networks over module_topology vfe, backbone_3d, map_to_bev_module,
    backbone_2d, dense_head (detector3d_template.py:22-51) + CenterPoint.forward (centerpoint.py:9-22),
    inference side. State-dict names equal the reference's (vfe., backbone_3d., backbone_2d., dense_head.)."""

    def __init__(self, model_cfg=None, num_class=3, class_names=("Vehicle", "Pedestrian", "Cyclist"),
                 point_cloud_range=(-75.2, -75.2, -2.0, 75.2, 75.2, 4.0), voxel_size=(0.1, 0.1, 0.15), num_point_features=5):
        super().__init__()
        cfg = model_cfg or waymo_centerpoint_cfg()
        self.model_cfg, self.num_class, self.class_names = cfg, num_class, list(class_names)
        self.point_cloud_range, self.voxel_size = list(point_cloud_range), list(voxel_size)
        self.num_point_features = num_point_features
        grid = ops.voxel_grid_size(self.voxel_size, self.point_cloud_range)[::-1]        # x,y,z like dataset.grid_size
        self.grid_size = np.array(grid)
        self.vfe = __all__[cfg.VFE.NAME](cfg.VFE, num_point_features=num_point_features, num_frames=1)
        self.backbone_3d = __all__[cfg.BACKBONE_3D.NAME](cfg.BACKBONE_3D, input_channels=num_point_features,
                                                         grid_size=self.grid_size, num_frames=1)
        self.map_to_bev_module = __all__[cfg.MAP_TO_BEV.NAME](cfg.MAP_TO_BEV)
        self.backbone_2d = __all__[cfg.BACKBONE_2D.NAME](cfg.BACKBONE_2D, num_frames=1, input_channels=cfg.MAP_TO_BEV.NUM_BEV_FEATURES)
        self.dense_head = __all__[cfg.DENSE_HEAD.NAME](cfg.DENSE_HEAD, num_frames=1, input_channels=self.backbone_2d.num_bev_features_post,
                                                       num_class=num_class, class_names=self.class_names, grid_size=self.grid_size,
                                                       point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size,
                                                       predict_boxes_when_training=False)
        self.module_list = [self.vfe, self.backbone_3d, self.map_to_bev_module, self.backbone_2d, self.dense_head]

    def forward(self, batch_dict):
        """centerpoint.py:9-22: training returns ({'loss': loss}, tb_dict, disp_dict) -- `loss.backward()` then reaches every
        parameter through the differentiable C-ABI convs (cpd_amd/autograd_ops.py); eval returns (pred_dicts, recall_dicts)."""
        batch_dict = self._run_modules(batch_dict)
        if self.training:
            loss_rpn, tb_dict = self.dense_head.get_loss()
            return {"loss": loss_rpn}, dict(loss_rpn=loss_rpn.item(), **tb_dict), {}
        return batch_dict["final_box_dicts"], {}

    FAST_EVAL_STICKY_STEPS = 16

    def _run_modules(self, batch_dict):
        """module_list over the batch_dict. Fast eval (spconv.install(fast_eval=True), every module in eval(), no autograd, f16x2): the
        modules run inside an optimistic range pass -- unscaled split-fp16 kernels, fp16-pair rows between the fused sparse layers,
        max |activation| recorded --, the verdict is read once behind the last module, and a step with an activation >= 2^15 is run
        AGAIN on the guarded kernels (exact at any magnitude); the model then stays guarded for FAST_EVAL_STICKY_STEPS steps, like
        the engine (engine.RANGE_STICKY_STEPS)."""
        fast = (not self.training and _spc.fast_eval() and default_conv_math() == "f16x2" and _fusable_eval(self)
                and getattr(self, "_guard_left", 0) <= 0)
        if fast:
            feats = batch_dict.get("voxel_features")
            fast = feats is not None and feats.is_cuda
        if fast:
            src = dict(batch_dict)
            with _spc.range_pass(feats.device) as rp:
                for m in self.module_list:
                    batch_dict = m(batch_dict)
                flag = rp.exceeded()
            if not int(flag.item()):
                return batch_dict
            import warnings
            warnings.warn("cpd_amd: an activation reached 2^15 -- step re-run with the range-guarded f16x2 kernels; the model stays "
                          "guarded for the next %d steps" % self.FAST_EVAL_STICKY_STEPS)
            self.range_reruns = getattr(self, "range_reruns", 0) + 1
            self._guard_left = self.FAST_EVAL_STICKY_STEPS + 1
            batch_dict = src
        if getattr(self, "_guard_left", 0) > 0:
            self._guard_left -= 1
        for m in self.module_list:
            batch_dict = m(batch_dict)
        return batch_dict

    def to_engine_config(self):
        from .engine import ModelConfig
        c = self.model_cfg
        if c.DENSE_HEAD.NAME != "CenterHead":            # anchor dense head: the backbone fields only (the head has its own config)
            return ModelConfig(point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size,
                               num_point_features=self.num_point_features, num_filters=list(c.BACKBONE_3D.NUM_FILTERS),
                               out_features=c.BACKBONE_3D.OUT_FEATURES, bev_layer_nums=list(c.BACKBONE_2D.LAYER_NUMS),
                               bev_layer_strides=list(c.BACKBONE_2D.LAYER_STRIDES), bev_num_filters=list(c.BACKBONE_2D.NUM_FILTERS),
                               bev_upsample_strides=list(c.BACKBONE_2D.UPSAMPLE_STRIDES),
                               bev_num_upsample_filters=list(c.BACKBONE_2D.NUM_UPSAMPLE_FILTERS), num_class=self.num_class)
        pp = c.DENSE_HEAD.POST_PROCESSING
        return ModelConfig(point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size,
                           num_point_features=self.num_point_features, num_filters=list(c.BACKBONE_3D.NUM_FILTERS),
                           out_features=c.BACKBONE_3D.OUT_FEATURES, bev_layer_nums=list(c.BACKBONE_2D.LAYER_NUMS),
                           bev_layer_strides=list(c.BACKBONE_2D.LAYER_STRIDES), bev_num_filters=list(c.BACKBONE_2D.NUM_FILTERS),
                           bev_upsample_strides=list(c.BACKBONE_2D.UPSAMPLE_STRIDES),
                           bev_num_upsample_filters=list(c.BACKBONE_2D.NUM_UPSAMPLE_FILTERS),
                           shared_conv_channel=c.DENSE_HEAD.SHARED_CONV_CHANNEL, num_class=self.num_class,
                           feature_map_stride=c.DENSE_HEAD.TARGET_ASSIGNER_CONFIG.FEATURE_MAP_STRIDE,
                           score_thresh=pp.SCORE_THRESH, post_center_limit_range=list(pp.POST_CENTER_LIMIT_RANGE),
                           max_obj_per_sample=pp.MAX_OBJ_PER_SAMPLE, nms_thresh=pp.NMS_CONFIG.NMS_THRESH,
                           nms_pre_maxsize=pp.NMS_CONFIG.NMS_PRE_MAXSIZE, nms_post_maxsize=pp.NMS_CONFIG.NMS_POST_MAXSIZE)

    def to_trainer(self, device="cuda", **kw):
        """The train step (hand-written backward + flat-buffer Adam) on this model's weights."""
        from .train_engine import CenterPointTrainer
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        return CenterPointTrainer(self.to_engine_config(), sd, device=device, **kw)

    def to_engine(self, device="cuda"):
        """The fused inference engine on this model's weights (eval-mode BatchNorm folded)."""
        from .engine import CenterPointEngine
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        return CenterPointEngine(self.to_engine_config(), sd, device=device)


def waymo_voxel_rcnn_cfg():
    """MODEL section of voxel_rcnn_cproto_center.yaml:13-183: the two-stage detector the shipped CPD config selects (`NAME: VoxelRCNN`):
    the CenterPoint first stage + ROI_HEAD (VoxelRCNNProtoHead, class-agnostic) + POST_PROCESSING (class-agnostic NMS 0.3)."""
    cfg = waymo_centerpoint_cfg()
    cfg.NAME = "VoxelRCNN"
    cfg.BACKBONE_3D.MM = True
    pool = dict(FEATURES_SOURCE=["x_conv3", "x_conv4"], PRE_MLP=True, GRID_SIZE=6,
                POOL_LAYERS={"x_conv3": dict(MLPS=[[32, 32], [32, 32]], QUERY_RANGES=[[2, 2, 2], [4, 4, 4]], POOL_RADIUS=[0.4, 0.8],
                                             NSAMPLE=[16, 16], POOL_METHOD="max_pool"),
                             "x_conv4": dict(MLPS=[[32, 32], [32, 32]], QUERY_RANGES=[[2, 2, 2], [4, 4, 4]], POOL_RADIUS=[0.8, 1.6],
                                             NSAMPLE=[16, 16], POOL_METHOD="max_pool")})
    import copy
    cfg.ROI_HEAD = AttrDict(
        NAME="VoxelRCNNProtoHead", CLASS_AGNOSTIC=True, SHARED_FC=[256, 256], CLS_FC=[256, 256], REG_FC=[256, 256], DP_RATIO=0.3,
        NMS_CONFIG=dict(TRAIN=dict(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=4000, NMS_POST_MAXSIZE=500, NMS_THRESH=0.8),
                        TEST=dict(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, USE_FAST_NMS=True, SCORE_THRESH=0.0, NMS_PRE_MAXSIZE=4000,
                                  NMS_POST_MAXSIZE=200, NMS_THRESH=0.8)),
        ROI_GRID_POOL=pool, ROI_GRID_POOL_PROTO=copy.deepcopy(pool),
        TARGET_CONFIG=dict(BOX_CODER="ResidualCoder", ROI_PER_IMAGE=130, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE="roi_iou",
                           CLS_FG_THRESH=0.6, CLS_BG_THRESH=0.02, CLS_BG_THRESH_LO=0.01, HARD_BG_RATIO=0.1, REG_FG_THRESH=0.3),
        LOSS_CONFIG=dict(CLS_LOSS="BinaryCrossEntropy", REG_LOSS="smooth-l1", CORNER_LOSS_REGULARIZATION=True, GRID_3D_IOU_LOSS=False,
                         LOSS_WEIGHTS=dict(rcnn_proto_weight=1.0, rcnn_cls_weight=1.0, rcnn_reg_weight=1.0, rcnn_corner_weight=1.0,
                                           rcnn_iou3d_weight=1.0, code_weights=[1.0] * 7)))
    cfg.POST_PROCESSING = AttrDict(RECALL_THRESH_LIST=[0.3, 0.5, 0.7], SCORE_THRESH=0.01, OUTPUT_RAW_SCORE=False, EVAL_METRIC="waymo",
                                   NMS_CONFIG=dict(MULTI_CLASSES_NMS=False, NMS_TYPE="nms_gpu", NMS_THRESH=0.3, NMS_PRE_MAXSIZE=4096,
                                                   NMS_POST_MAXSIZE=50075))
    return cfg


def waymo_voxel_rcnn_dbscan_cfg():
    """MODEL section of voxel_rcnn_dbscan_single_train.yaml:12-174 (voxel_rcnn_oyster_single_train.yaml is the same model): `VoxelRCNN` =
    AnchorHeadSingleV2 proposals -> VoxelRCNNHead (class-agnostic) -> class-agnostic NMS 0.3 at score 0.1."""
    from .anchor_engine import dbscan_dense_head_cfg
    cfg = waymo_voxel_rcnn_cfg()
    cfg.DENSE_HEAD = AttrDict(dbscan_dense_head_cfg())
    cfg.BACKBONE_3D.MM = False
    rh = cfg.ROI_HEAD
    rh.NAME = "VoxelRCNNHead"
    rh.NMS_CONFIG["TRAIN"]["NMS_POST_MAXSIZE"] = 500
    rh.NMS_CONFIG["TEST"]["NMS_POST_MAXSIZE"] = 200
    rh.TARGET_CONFIG.update(ROI_PER_IMAGE=150, CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8, REG_FG_THRESH=0.55)
    cfg.POST_PROCESSING.SCORE_THRESH = 0.1
    return cfg


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.class_agnostic_nms (cpd/models/model_utils/model_nms_utils.py:113-134) on the B3 operators."""
    src = box_scores
    if score_thresh is not None:
        mask = box_scores >= score_thresh
        box_scores, box_preds = box_scores[mask], box_preds[mask]
    selected = box_scores.new_zeros((0,), dtype=torch.long)
    if box_scores.shape[0] > 0:
        # (the reference's torch.topk leaves the order of EQUAL scores to the implementation; here, and in the fused engine, ties go
        # to the lower index -- a stable descending sort -- so that both paths keep the same boxes among tied, overlapping ones)
        ranked, order = torch.sort(box_scores, descending=True, stable=True)
        k = min(nms_config["NMS_PRE_MAXSIZE"], box_scores.shape[0])
        top, order = ranked[:k], order[:k]
        fn = getattr(iou3d_nms_utils, nms_config["NMS_TYPE"])
        keep, _ = fn(box_preds[order][:, 0:7], top, nms_config["NMS_THRESH"])
        selected = order[keep[:nms_config["NMS_POST_MAXSIZE"]]]
    if score_thresh is not None:
        selected = mask.nonzero().view(-1)[selected]
    return selected, src[selected]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """model_nms_utils.multi_classes_nms (cpd/models/model_utils/model_nms_utils.py:137-170) on the B3 operators: per class column k --
    score threshold, the NMS_PRE_MAXSIZE best, rotated NMS, the first NMS_POST_MAXSIZE -- concatenated class by class.
    -> (pred_scores, pred_labels (the column index k, 0-based as in the reference), pred_boxes). Ties rank the lower index first (a
    stable sort, like class_agnostic_nms here); with score_thresh None every row competes (the reference reads an unset variable there)."""
    pred_scores, pred_labels, pred_boxes = [], [], []
    for k in range(cls_scores.shape[1]):
        if score_thresh is not None:
            mask = cls_scores[:, k] >= score_thresh
            box_scores, cur = cls_scores[mask, k], box_preds[mask]
        else:
            box_scores, cur = cls_scores[:, k], box_preds
        selected = box_scores.new_zeros((0,), dtype=torch.long)
        if box_scores.shape[0] > 0:
            ranked, order = torch.sort(box_scores, descending=True, stable=True)
            n = min(nms_config["NMS_PRE_MAXSIZE"], box_scores.shape[0])
            top, order = ranked[:n], order[:n]
            keep, _ = getattr(iou3d_nms_utils, nms_config["NMS_TYPE"])(cur[order][:, 0:7], top, nms_config["NMS_THRESH"])
            selected = order[keep[:nms_config["NMS_POST_MAXSIZE"]]]
        pred_scores.append(box_scores[selected])
        pred_labels.append(torch.full((len(selected),), k, dtype=torch.long, device=box_scores.device))
        pred_boxes.append(cur[selected])
    return torch.cat(pred_scores, dim=0), torch.cat(pred_labels, dim=0), torch.cat(pred_boxes, dim=0)


def post_process_frame(pp, num_class, box_preds, cls_preds, normalized, label_preds=None, label_mapping=None):
    """One frame of Detector3DTemplate.post_processing (cpd/models/detectors/detector3d_template.py:246-331): every branch of it --
    MULTI_CLASSES_NMS (single tensor or the multi-head list with `label_mapping`), WBF (score mask only: the fusion itself happens in the
    evaluation scripts), class-agnostic NMS with OUTPUT_RAW_SCORE. `label_preds`: the RoI / dense-head labels when has_class_labels.
    -> (boxes, scores, labels)"""
    multi = isinstance(cls_preds, (list, tuple))
    src = cls_preds
    if not normalized:
        cls_preds = [torch.sigmoid(x) for x in cls_preds] if multi else torch.sigmoid(cls_preds)
    nms = pp.NMS_CONFIG
    if nms["MULTI_CLASSES_NMS"]:
        if not multi:
            cls_preds = [cls_preds]
            # (the reference builds arange(1, num_class) here -- one entry short of its own assert on the next line; classes 1..num_class)
            label_mapping = [torch.arange(1, cls_preds[0].shape[1] + 1, device=cls_preds[0].device)]
        start, scores, labels, boxes = 0, [], [], []
        for cur, mapping in zip(cls_preds, label_mapping):
            assert cur.shape[1] == len(mapping)
            s_, l_, b_ = multi_classes_nms(cur, box_preds[start:start + cur.shape[0]], nms, pp.SCORE_THRESH)
            scores.append(s_); labels.append(mapping[l_]); boxes.append(b_)
            start += cur.shape[0]
        return torch.cat(boxes, dim=0), torch.cat(scores, dim=0), torch.cat(labels, dim=0)
    assert not multi, "a multi-head class list needs MULTI_CLASSES_NMS (detector3d_template.py:268-290)"
    cls, labels = torch.max(cls_preds, dim=-1)
    labels = label_preds if label_preds is not None else labels + 1
    if pp.get("WBF", False):
        mask = cls > pp.SCORE_THRESH
        return box_preds[mask], cls[mask], labels[mask]
    sel, scores = class_agnostic_nms(cls, box_preds, nms, pp.SCORE_THRESH)
    if pp.OUTPUT_RAW_SCORE:
        scores = torch.max(src, dim=-1)[0][sel]
    return box_preds[sel], scores, labels[sel]


class VoxelRCNN(CenterPoint):
    """The two-stage detector (cpd/models/detectors/voxel_rcnn.py:3-43 over Detector3DTemplate's module_topology with `roi_head`,
    detector3d_template.py:22-25,170-190): CenterPoint's modules with the dense head predicting boxes in every mode (its NMS output is
    the second stage's `rois`, center_head.py:340-350), the RoI head of ROI_HEAD.NAME, and `post_processing`
    (detector3d_template.py:222-343: sigmoid, the RoI labels, class-agnostic NMS at POST_PROCESSING.NMS_CONFIG). Training returns
    ({'loss': loss_rpn + loss_rcnn}, tb_dict, disp_dict); eval (pred_dicts, recall_dicts, batch_dict) as the reference does.
    State-dict prefixes: vfe. backbone_3d. backbone_2d. dense_head. roi_head."""

    def __init__(self, model_cfg=None, num_class=3, **kw):
        cfg = model_cfg or waymo_voxel_rcnn_cfg()
        super().__init__(cfg, num_class=num_class, **kw)
        self.dense_head.predict_boxes_when_training = True
        rc = cfg.ROI_HEAD
        self.roi_head = __all__[rc.NAME](input_channels=self.backbone_3d.num_point_features, model_cfg=rc,
                                         point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size,
                                         num_class=1 if rc.get("CLASS_AGNOSTIC", False) else num_class)
        self.module_list = self.module_list + [self.roi_head]

    def post_processing(self, batch_dict):
        pp = self.model_cfg.POST_PROCESSING
        pred_dicts = []
        for b in range(batch_dict["batch_size"]):
            boxes = batch_dict["batch_box_preds"][b]
            src = batch_dict["batch_cls_preds"]
            src = [x[b] for x in src] if isinstance(src, (list, tuple)) else src[b]
            if not isinstance(src, list):
                assert src.shape[1] in (1, self.num_class)
            labels = None
            if batch_dict.get("has_class_labels", False):
                labels = batch_dict["roi_labels" if "roi_labels" in batch_dict else "batch_pred_labels"][b]
            fb, fs, fl = post_process_frame(pp, self.num_class, boxes, src, batch_dict["cls_preds_normalized"], labels,
                                            batch_dict.get("multihead_label_mapping"))
            rec = {"pred_boxes": fb, "pred_scores": fs, "pred_labels": fl}
            if pp.get("WBF", False):
                rec["WBF"] = True
            pred_dicts.append(rec)
        return pred_dicts, {}

    def forward(self, batch_dict):
        batch_dict = self._run_modules(batch_dict)          # (fast eval: the first stage inside the optimistic range pass, CenterPoint._run_modules;
        if self.training:                                   # the RoI head reads the pooled levels' `.features` -- fp32 rows, decoded on demand)
            loss_rpn, tb_dict = self.dense_head.get_loss()
            loss_rcnn, tb_dict = self.roi_head.get_loss(tb_dict)
            return {"loss": loss_rpn + loss_rcnn}, tb_dict, {}
        pred_dicts, recall = self.post_processing(batch_dict)
        return pred_dicts, recall, batch_dict

    def to_engine(self, device="cuda", **kw):
        """The fused two-stage inference engine (cpd_amd/two_stage.py) on this model's weights; with an anchor dense head
        (voxel_rcnn_dbscan / oyster configs) its first stage is cpd_amd.anchor_engine.AnchorPointEngine."""
        from .two_stage import VoxelRCNNEngine
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        rpn = None
        if self.model_cfg.DENSE_HEAD.NAME != "CenterHead":
            from .anchor_engine import AnchorPointEngine
            rpn = AnchorPointEngine(self.to_engine_config(), sd, self.model_cfg.DENSE_HEAD, self.model_cfg.ROI_HEAD.NMS_CONFIG["TEST"],
                                    class_names=self.class_names, device=device)
        return VoxelRCNNEngine(self.to_engine_config(), self.model_cfg.ROI_HEAD, self.model_cfg.POST_PROCESSING, sd, device=device, rpn=rpn, **kw)
