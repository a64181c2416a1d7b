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
