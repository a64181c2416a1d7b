"""Drop-in API checks that need no GPU: the mirrored modules expose the reference's state_dict names
and layouts, the spconv shim registers the names the reference imports, registries resolve by NAME."""
import sys

import torch

from cpd_amd import models
from cpd_amd.engine import ModelConfig, init_state_dict


def test_state_dict_names_match_reference_layout():
    net = models.CenterPoint()
    sd = net.state_dict()
    ref = init_state_dict(ModelConfig(), 0)
    assert set(sd.keys()) == set(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    # spconv-2.x weight layout (Cout, kD, kH, kW, Cin) -- detector3d_template.py:388-419
    assert tuple(sd["backbone_3d.conv_input.0.weight"].shape) == (16, 3, 3, 3, 5)
    assert tuple(sd["backbone_3d.conv_out.0.weight"].shape) == (128, 3, 1, 1, 128)
    assert tuple(sd["backbone_2d.deblocks.1.0.weight"].shape) == (256, 256, 2, 2)
    net.load_state_dict(ref)           # a reference-named checkpoint loads unchanged


def test_spconv_shim_names():
    import cpd_amd.spconv as shim
    shim.install()
    import spconv.pytorch as spconv
    from spconv.pytorch.utils import PointToVoxel, gather_features_by_pc_voxel_id  # noqa: F401
    from spconv.utils import Point2VoxelCPU3d  # noqa: F401
    import cumm.tensorview as tv
    assert hasattr(tv, "from_numpy")
    for n in ["SparseConvTensor", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d", "SparseSequential", "SparseModule"]:
        assert hasattr(spconv, n)
    conv = spconv.SubMConv3d(4, 8, 3, padding=1, bias=False, indice_key="k")
    assert isinstance(conv, spconv.conv.SparseConvolution) and conv.bias is None
    assert tuple(conv.weight.shape) == (8, 3, 3, 3, 4)
    seq = spconv.SparseSequential(conv, torch.nn.BatchNorm1d(8), torch.nn.ReLU())
    assert list(seq.state_dict())[0] == "0.weight"
    for m in ("spconv", "spconv.pytorch", "spconv.pytorch.utils", "spconv.utils", "cumm.tensorview"):
        assert m in sys.modules


def test_registry_and_cfg():
    cfg = models.waymo_centerpoint_cfg()
    assert models.__all__[cfg.BACKBONE_3D.NAME] is models.VoxelResBackBone8x
    net = models.CenterPoint(cfg)
    assert net.backbone_3d.sparse_shape == [41, 1504, 1504]
    assert net.backbone_2d.num_bev_features_post == 512
    ecfg = net.to_engine_config()
    assert ecfg.sparse_shape == [41, 1504, 1504] and ecfg.nms_thresh == 0.8


def test_center_head_training_branch_targets_and_loss():
    """CenterHead.assign_targets / get_loss (center_head.py:159-250) through the module API, on CPU:
    same numbers as cpd_amd.center_loss (itself pinned on the reference goldens)."""
    from cpd_amd import center_loss
    cfg = models.waymo_centerpoint_cfg()
    net = models.CenterPoint(cfg)
    head = net.dense_head
    torch.manual_seed(0)
    B, h, w = 2, 24, 24
    gt = torch.zeros(B, 6, 8)
    for b in range(B):
        for i in range(4):
            gt[b, i] = torch.tensor([float(torch.empty(1).uniform_(-9, 9)), float(torch.empty(1).uniform_(-9, 9)), 0.8, 4.2, 1.9,
                                     1.6, 0.3 * i, float(1 + (i % 3))])
    head.point_cloud_range = [-9.6, -9.6, -2.0, 9.6, 9.6, 4.0]
    td = head.assign_targets(gt, feature_map_size=(h, w))
    heat, tgt, inds, masks = center_loss.assign_targets(gt, (h, w), head.point_cloud_range, head.voxel_size, 3, 8)
    assert torch.equal(td["heatmaps"][0], heat) and torch.equal(td["inds"][0], inds) and torch.equal(td["masks"][0], masks)
    assert int(masks.sum()) == 8
    pd = {"hm": torch.randn(B, 3, h, w), "center": torch.randn(B, 2, h, w), "center_z": torch.randn(B, 1, h, w),
          "dim": torch.randn(B, 3, h, w), "rot": torch.randn(B, 2, h, w)}
    head.forward_ret_dict = {"pred_dicts": [pd], "target_dicts": td}
    loss, tb = head.get_loss()
    rows = torch.cat([pd[k] for k in ("center", "center_z", "dim", "rot", "hm")], 1).permute(0, 2, 3, 1).reshape(B * h * w, -1)
    want, _ = center_loss.center_head_loss(rows, B, h, w, heat, tgt, inds, masks, 3, hm_col=8)
    assert abs(float(loss) - float(want)) < 1e-6 and abs(tb["rpn_loss"] - float(want)) < 1e-6


def test_mm_branch_layers_follow_the_reference_names():
    """BACKBONE_3D.MM (voxel_rcnn_cproto_center.yaml:23): the prototype branch's encoder, spconv_backbone.py:456-486 --
    conv_input_2, conv1_2 (two residual blocks), conv{2,3,4}_2 (strided conv + ONE residual block)."""
    cfg = models.waymo_centerpoint_cfg()
    cfg.BACKBONE_3D.MM = True
    bb = models.VoxelResBackBone8x(cfg.BACKBONE_3D, input_channels=5, grid_size=[1504, 1504, 40])
    keys = set(bb.state_dict())
    assert {"conv_input_2.0.weight", "conv1_2.1.conv2.weight", "conv2_2.0.0.weight", "conv2_2.1.bn2.running_var", "conv4_2.1.conv1.bias"} <= keys
    assert not any(k.startswith("conv2_2.2.") for k in keys)            # one residual block per strided level
    assert tuple(bb.state_dict()["conv4_2.0.0.weight"].shape) == (128, 3, 3, 3, 64)
    plain = models.VoxelResBackBone8x(models.waymo_centerpoint_cfg().BACKBONE_3D, input_channels=5, grid_size=[1504, 1504, 40])
    assert not any("_2." in k for k in plain.state_dict())


def test_fused_eval_needs_the_whole_tree_in_eval_mode():
    """a container in eval() whose BatchNorm was switched back to .train() (or has no running statistics) must not take
    the fused path -- it would fold running statistics the module is not using."""
    from cpd_amd.spconv.pytorch.conv import fusable_eval
    seq = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1), torch.nn.BatchNorm2d(4), torch.nn.ReLU()).eval()
    with torch.no_grad():
        assert fusable_eval(seq)
        seq[1].train()
        assert not seq.training and not fusable_eval(seq)          # container still in eval, its BatchNorm is not
        seq[1].eval()
        assert fusable_eval(seq)
        nostats = torch.nn.Sequential(torch.nn.BatchNorm2d(4, track_running_stats=False)).eval()
        assert not fusable_eval(nostats)
    assert not fusable_eval(seq)                                   # autograd on: never fused


def test_range_tag_is_retired_by_an_in_place_write():
    """the f16x2 range block rides on the tensor as an attribute; an in-place op afterwards makes it stale."""
    from cpd_amd import ops
    t = torch.ones(8, 32)
    blk = torch.zeros(ops.ABSMAX_WORDS, dtype=torch.int32)
    ops.tag_range(t, blk)
    assert ops.tagged_range(t) is blk
    v = t.view(4, 64)                       # a view shares the version counter
    v.mul_(1e6)
    assert ops.tagged_range(t) is None      # stale: the consumer measures the tensor again
    assert ops.tagged_range(torch.ones(2, 2)) is None


def test_epilogue_shift_with_scale_but_no_shift_keeps_the_bias_inside_the_scale():
    """acc * scale + shift with shift omitted must mean (acc + bias) * scale."""
    from cpd_amd import ops
    bias, scale = torch.tensor([1.0, -2.0]), torch.tensor([3.0, 0.5])
    assert torch.equal(ops.epilogue_shift(scale, None, bias), bias * scale)
    assert torch.equal(ops.epilogue_shift(None, None, bias), bias)
    given = torch.tensor([7.0, 7.0])
    assert ops.epilogue_shift(scale, given, bias) is given
    assert ops.epilogue_shift(scale, None, None) is None


def test_voxel_backbone8x_is_registered_with_the_reference_parameter_names():
    """spconv_backbone.py:138-232 / backbones_3d/__init__.py:3-8: the non-residual backbone, same module tree (so its checkpoints load)."""
    cfg = models.waymo_centerpoint_cfg().BACKBONE_3D
    bb = models.__all__["VoxelBackBone8x"](cfg, input_channels=5, grid_size=[1504, 1504, 40], num_frames=1)
    sd = bb.state_dict()
    convs = sorted(k for k in sd if sd[k].dim() == 5)
    assert convs == sorted(["conv_input.0.weight", "conv1.0.0.weight", "conv_out.0.weight"] +
                           ["conv%d.%d.0.weight" % (s, i) for s in (2, 3, 4) for i in (0, 1, 2)])
    assert tuple(sd["conv2.0.0.weight"].shape) == (32, 3, 3, 3, 16) and tuple(sd["conv_out.0.weight"].shape) == (128, 3, 1, 1, 128)
    assert not any(k.endswith(".bias") and sd[k].dim() == 1 and ".0.bias" in k for k in sd)      # post_act_block convs carry no bias
    assert bb.sparse_shape == [41, 1504, 1504] and bb.num_point_features == {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64, "x_conv4": 128}
    cfg.MM = True
    mm = models.VoxelBackBone8x(cfg, input_channels=5, grid_size=[400, 400, 40])
    assert "conv4_2.2.0.weight" in mm.state_dict() and "conv_input_2.0.weight" in mm.state_dict()


def test_pair_rows_travel_as_a_type_and_decode_on_demand():
    """ops.PairRows (round 6): the fp16-pair layout travels as a TYPE, not as an attribute a .contiguous() would drop; a
    SparseConvTensor that carries pair rows decodes `.features` once, exactly (x = h + l for values with <= 22 significant bits), and a
    plain assignment to `.features` retires the pairs."""
    from cpd_amd import ops
    from cpd_amd.spconv.pytorch.core import SparseConvTensor
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50, 64, generator=g)
    x = x.half().float() + (torch.randn(50, 64, generator=g) * 1e-4).half().float()      # exactly representable as h + l
    pairs = ops.rows_to_pairs(x)
    p = ops.PairRows(pairs)
    assert tuple(p.shape) == (50, 64) and torch.equal(p.float_rows(), x)
    x16 = x[:, :16].contiguous()
    assert torch.equal(ops.PairRows(ops.rows_to_pairs(x16)).float_rows(), x16)            # the 16-channel [hi 4 | lo 4] form
    for bad in (pairs[:, :48], pairs.t(), pairs.double()):                                # not whole blocks / not contiguous / not fp32-typed
        try:
            ops.PairRows(bad)
            raise RuntimeError("accepted")
        except AssertionError:
            pass
    idx = torch.zeros((50, 4), dtype=torch.int32)
    t = SparseConvTensor(p, idx, [4, 8, 8], 1)
    assert t._pairs is p and t._features is None
    f = t.features
    assert torch.equal(f, x) and t.features is f                                          # decoded once, kept
    t2 = t.replace_feature(p)
    assert t2._pairs is p and t2.indice_dict is t.indice_dict
    t.features = x * 2
    assert t._pairs is None and torch.equal(t.features, x * 2)


def test_range_pass_bookkeeping_and_fast_eval_switches():
    """spconv.pytorch.conv.range_pass: blocks are handed out of a zeroed pool inside the pass only, the verdict is the pool's maximum against
    2^15, the state is per thread, and install(fast_eval=...) / set_fast_eval toggle the opt-in."""
    import threading
    import cpd_amd.spconv as shim
    from cpd_amd.spconv.pytorch import conv as spc
    old = spc.fast_eval()
    try:
        shim.install(fast_eval=True)
        assert spc.fast_eval()
        spc.set_fast_eval(False)
        assert not spc.fast_eval() and not spc.optimistic()
        with spc.range_pass("cpu", n_blocks=4) as rp:
            assert spc.optimistic()
            a, b = spc.record_block(), spc.record_block()
            assert a.data_ptr() != b.data_ptr() and int(a.abs().max()) == 0
            seen = []
            th = threading.Thread(target=lambda: seen.append(spc.optimistic()))           # another thread is not inside this pass
            th.start(); th.join()
            assert seen == [False]
            assert int(rp.exceeded()) == 0
            b[0] = 0x47000000                                                             # bits of 32768.0f
            assert int(rp.exceeded()) == 1
            spc.record_block(); spc.record_block()
            try:
                spc.record_block()
                raise RuntimeError("a fifth block of four")
            except RuntimeError as e:
                assert "range_pass" in str(e)
        assert not spc.optimistic()
        with spc.range_pass("cpu", n_blocks=4) as rp:                                     # the pool is re-zeroed
            assert int(rp.exceeded()) == 0
    finally:
        spc.set_fast_eval(old)
