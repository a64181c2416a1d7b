"""End-to-end parity: the fused HIP engine vs the reference module graph on the CPU oracle
(tests/ref_pipeline.py), stage by stage, on a reduced geometry the oracle finishes in seconds;
plus size-independent properties at the full Waymo-shape size of BASELINE.json's config 2."""
import numpy as np
import pytest
import torch

from cpd_amd import ops
from cpd_amd.engine import CenterPointEngine, ModelConfig, init_state_dict
from cpd_amd.synthetic import waymo_cloud

import ref_pipeline

pytestmark = pytest.mark.gpu


def small_cfg():
    # 40 m x 40 m crop of the Waymo geometry: grid 400x400x40 -> BEV 50x50, narrower BEV widths
    return ModelConfig(point_cloud_range=[-20.0, -20.0, -2.0, 20.0, 20.0, 4.0], post_center_limit_range=[-20, -20, -2, 20, 20, 4],
                       bev_num_filters=[64, 128], bev_num_upsample_filters=[128, 128], bev_layer_nums=[2, 2],
                       max_obj_per_sample=100)


def canon(idx):
    return np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))


def test_engine_matches_reference_graph(oracle, hip):
    cfg = small_cfg()
    sd = init_state_dict(cfg, seed=3)
    pts = [waymo_cloud(0, n_points=60000), waymo_cloud(1, n_points=50000)]
    for p in pts:
        p[:, :2] *= 0.3                                    # pull the scene into the crop
    eng = CenterPointEngine(cfg, sd)
    res, it = eng.forward([torch.from_numpy(p).cuda() for p in pts], return_intermediates=True)
    ref, rt = ref_pipeline.forward(oracle, cfg, sd, pts)

    # the engine keeps its rows in its own order (level 0: canonical instead of first appearance, ModelConfig.voxel_row_order;
    # strided levels: tap-pattern order above row_order_min_rows): voxel list and levels are compared as sets, both sides sorted
    shape0 = cfg.sparse_shape
    vf, vc, _ = _canonical(it["voxel_features"], it["voxel_coords"], shape0)
    vf0, vc0, _ = _canonical(rt["voxel_features"], rt["voxel_coords"], shape0)
    np.testing.assert_array_equal(vc, vc0)
    np.testing.assert_allclose(vf, vf0, rtol=1e-6, atol=1e-6)
    for name in ["x_conv1", "x_conv2", "x_conv3", "x_conv4"]:
        f, i, s = it["levels"][name]
        f0, i0, s0 = rt["levels"][name]
        assert list(s) == list(s0)
        f, i, _ = _canonical(f, i, s)
        f0, i0, _ = _canonical(f0, i0, s0)
        np.testing.assert_array_equal(i, i0)
        np.testing.assert_allclose(f, f0, atol=1e-4, rtol=0, err_msg=name)
    x, idx, shape = it["encoded"]
    x0, idx0, shape0 = rt["encoded"]
    np.testing.assert_array_equal(idx.cpu().numpy(), idx0)
    np.testing.assert_allclose(x.cpu().numpy(), x0, atol=1e-4, rtol=0)
    d, h, w = shape
    B = len(pts)
    # spatial_features: ours is channels-last with channel z*C+c, reference (B, C*D, H, W) with c*D+z
    C = x.shape[1]
    ours = it["spatial_features_nhwc"].cpu().numpy().reshape(B, h, w, d, C).transpose(0, 4, 3, 1, 2).reshape(B, C * d, h, w)
    np.testing.assert_allclose(ours, rt["spatial_features"], atol=1e-4, rtol=0)
    bev = it["bev_cat"].cpu().numpy().reshape(B, h, w, -1).transpose(0, 3, 1, 2)
    np.testing.assert_allclose(bev, rt["bev"], atol=1e-4, rtol=0)
    head = it["head_rows"].cpu().numpy().reshape(B, h, w, -1).transpose(0, 3, 1, 2)
    for name, (c0, cn) in eng.head_slices.items():
        np.testing.assert_allclose(head[:, c0:c0 + cn], rt["heads"][name], atol=1e-4, rtol=0, err_msg=name)
    for b in range(B):
        got, want = res[b], ref[b]
        assert got["pred_boxes"].shape[0] == want["pred_boxes"].shape[0]
        np.testing.assert_array_equal(got["pred_labels"].cpu().numpy(), want["pred_labels"])
        np.testing.assert_allclose(got["pred_scores"].cpu().numpy(), want["pred_scores"], atol=1e-5)
        np.testing.assert_allclose(got["pred_boxes"].cpu().numpy(), want["pred_boxes"], atol=1e-3, rtol=1e-4)


def _canonical(f, i, shape):
    """rows of a level sorted by (b, z, y, x); third value: was the level stored in another order?"""
    i = i.cpu().numpy() if hasattr(i, "cpu") else np.asarray(i)
    f = f.cpu().numpy() if hasattr(f, "cpu") else np.asarray(f)
    key = ((i[:, 0].astype(np.int64) * shape[0] + i[:, 1]) * shape[1] + i[:, 2]) * shape[2] + i[:, 3]
    o = np.argsort(key, kind="stable")
    return f[o], i[o], bool((o != np.arange(len(o))).any())


@pytest.mark.parametrize("math", ["bf16x3", "f16x2"])
def test_full_size_config2_matches_oracle(oracle, hip, math):
    """BASELINE config 2 at FULL size, features included: one 160k-point W-cloud through the oracle's
    un-fused reference graph (seconds on the GPU box's host cores) against the engine on a batch of four frames -- three
    other clouds plus that frame -- so that every layer runs the kernel instantiation the benchmark runs (the batched
    voxelizer, 128-row row-wave tiles, the window / workgroup split kernels; a single frame would take the small-problem
    variants). Indices bit-exact; every sparse level, the BEV map and the head maps <= 1e-4."""
    cfg = ModelConfig(conv_math=math)
    sd = init_state_dict(cfg, seed=0)
    pts = waymo_cloud(0)
    ref, rt = ref_pipeline.forward(oracle, cfg, sd, [pts])
    eng = CenterPointEngine(cfg, sd)
    clouds = [torch.from_numpy(waymo_cloud(k)).cuda() for k in (1, 2, 3)] + [torch.from_numpy(pts).cuda()]
    fi = len(clouds) - 1                                           # the checked frame sits LAST: nonzero row offsets everywhere
    res, it = eng.forward(clouds, return_intermediates=True)

    vf, vc, _ = _canonical(it["voxel_features"], it["voxel_coords"], cfg.sparse_shape)
    vf0, vc0, _ = _canonical(rt["voxel_features"], rt["voxel_coords"], cfg.sparse_shape)
    mine = vc[:, 0] == fi
    np.testing.assert_array_equal(vc[mine][:, 1:], vc0[:, 1:])
    np.testing.assert_allclose(vf[mine], vf0, rtol=1e-6, atol=1e-6)
    reordered = 0
    for name in ["x_conv1", "x_conv2", "x_conv3", "x_conv4"]:
        f, i, s = it["levels"][name]
        f0, i0, s0 = rt["levels"][name]
        # the engine keeps the strided levels in tap-pattern order (ModelConfig.row_order): a level is a set of (site, feature)
        # pairs -- compared here in canonical (b, z, y, x) order, where the site LIST must equal the oracle's bit for bit
        f, i, moved = _canonical(f, i, s)
        f0, i0, _ = _canonical(f0, i0, s0)                          # (the oracle's level 0 is in first-appearance order)
        reordered += int(moved and name != "x_conv1")
        mine = i[:, 0] == fi
        np.testing.assert_array_equal(i[mine][:, 1:], i0[:, 1:])
        np.testing.assert_allclose(f[mine], f0, atol=1e-4, rtol=0, err_msg=name)
    assert reordered == 3                                           # x_conv2..4 really ran in the bench's row order
    x, idx, shape = it["encoded"]
    x0, idx0, shape0 = rt["encoded"]
    idx = idx.cpu().numpy()
    mine = idx[:, 0] == fi
    assert list(shape) == list(shape0) == [2, 188, 188]
    np.testing.assert_array_equal(idx[mine][:, 1:], idx0[:, 1:])
    np.testing.assert_allclose(x.cpu().numpy()[mine], x0, atol=1e-4, rtol=0)
    d, h, w = shape
    B = len(clouds)
    bev = it["bev_cat"].view(B, h, w, -1)[fi].permute(2, 0, 1).cpu().numpy()
    np.testing.assert_allclose(bev, rt["bev"][0], atol=1e-4, rtol=0)
    head = it["head_rows"].view(B, h, w, -1)[fi].permute(2, 0, 1).cpu().numpy()
    for name, (c0, cn) in eng.head_slices.items():
        np.testing.assert_allclose(head[c0:c0 + cn], rt["heads"][name][0], atol=1e-4, rtol=0, err_msg=name)
    got, want = res[fi], ref[0]
    a, b = got["pred_boxes"].cpu().numpy(), want["pred_boxes"]
    assert a.shape == b.shape
    np.testing.assert_allclose(got["pred_scores"].cpu().numpy(), want["pred_scores"], atol=1e-5)
    # same detections; boxes whose scores tie to ~1e-6 may swap ranks between the two pipelines, so match by geometry
    j = np.abs(a[:, None, :] - b[None, :, :]).max(-1).argmin(1)
    assert sorted(j.tolist()) == list(range(len(b)))
    np.testing.assert_allclose(a, b[j], atol=1e-3, rtol=1e-4)
    np.testing.assert_array_equal(got["pred_labels"].cpu().numpy(), want["pred_labels"][j])
    assert np.abs(j - np.arange(len(j))).max() <= 3           # only neighbours in the ranking swap


def test_full_size_properties(hip):
    """Config 2 (160k Waymo-shape cloud, full widths): shapes, determinism, batch consistency."""
    cfg = ModelConfig()
    sd = init_state_dict(cfg, seed=0)
    eng = CenterPointEngine(cfg, sd)
    p0 = torch.from_numpy(waymo_cloud(0)).cuda()
    p1 = torch.from_numpy(waymo_cloud(1)).cuda()
    r0, it0 = eng.forward([p0], return_intermediates=True)
    assert it0["encoded"][2] == [2, 188, 188]
    assert it0["bev_cat"].shape == (188 * 188, 512)
    assert torch.isfinite(it0["head_rows"][:, :11]).all()
    s = r0[0]["pred_scores"].cpu().numpy()
    assert (np.diff(s) <= 0).all() and r0[0]["pred_boxes"].shape[0] <= cfg.nms_post_maxsize
    # determinism (no atomics in the numeric path): a second run is bit-identical
    r0b, it0b = eng.forward([p0], return_intermediates=True)
    assert torch.equal(it0["head_rows"][:, :11], it0b["head_rows"][:, :11])
    assert torch.equal(r0[0]["pred_boxes"], r0b[0]["pred_boxes"])
    # a frame's result does not depend on what else is in the batch (different batch sizes run different
    # kernel instantiations -- fp32-MFMA vs split-bf16 tiles -- so equality is to fp32 rounding, not bitwise)
    rb = eng.forward([p1, p0])
    a, b = rb[1]["pred_boxes"].cpu().numpy(), r0[0]["pred_boxes"].cpu().numpy()
    assert a.shape == b.shape
    # same detections; boxes whose scores tie to ~1e-6 may swap ranks between kernel paths, so match by geometry
    d = np.abs(a[:, None, :] - b[None, :, :]).max(-1)
    j = d.argmin(1)
    assert sorted(j.tolist()) == list(range(len(b)))
    np.testing.assert_allclose(a, b[j], atol=1e-4, rtol=2e-5)
    np.testing.assert_array_equal(rb[1]["pred_labels"].cpu().numpy(), r0[0]["pred_labels"].cpu().numpy()[j])
    np.testing.assert_allclose(rb[1]["pred_scores"].cpu().numpy(), r0[0]["pred_scores"].cpu().numpy()[j], atol=2e-5)


def test_module_api_matches_engine(oracle, hip):
    """The un-fused drop-in modules (cpd_amd.models, reference batch_dict contract) and the fused
    engine give the same detections from the same reference-named weights; the numpy
    VoxelGeneratorWrapper mirror reproduces the oracle's voxels."""
    from cpd_amd import models
    from cpd_amd.voxel_generator import VoxelGeneratorWrapper
    cfg = small_cfg()
    cfg.bev_layer_nums = [5, 5]
    mcfg = models.waymo_centerpoint_cfg()
    mcfg.BACKBONE_2D.NUM_FILTERS = cfg.bev_num_filters
    mcfg.BACKBONE_2D.NUM_UPSAMPLE_FILTERS = cfg.bev_num_upsample_filters
    mcfg.DENSE_HEAD.POST_PROCESSING.POST_CENTER_LIMIT_RANGE = cfg.post_center_limit_range
    mcfg.DENSE_HEAD.POST_PROCESSING.MAX_OBJ_PER_SAMPLE = cfg.max_obj_per_sample
    net = models.CenterPoint(mcfg, point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size).cuda().eval()
    sd = init_state_dict(cfg, seed=5)
    net.load_state_dict(sd)
    pts = waymo_cloud(2, n_points=40000)
    pts[:, :2] *= 0.3
    gen = VoxelGeneratorWrapper(cfg.voxel_size, cfg.point_cloud_range, 5, cfg.max_points_per_voxel, cfg.max_voxels)
    voxels, coords, num = gen.generate(pts)
    v0, c0, n0 = oracle.voxelize(pts, cfg.voxel_size, cfg.point_cloud_range, cfg.max_points_per_voxel, cfg.max_voxels)
    np.testing.assert_array_equal(coords, c0); np.testing.assert_array_equal(voxels, v0); np.testing.assert_array_equal(num, n0)
    # collate_batch pad + load_data_to_gpu's float cast (dataset.py:264, models/__init__.py:24)
    batch = {"voxels": torch.from_numpy(voxels).cuda(), "voxel_num_points": torch.from_numpy(num).float().cuda(),
             "voxel_coords": torch.from_numpy(np.pad(coords, ((0, 0), (1, 0)))).float().cuda(), "batch_size": 1}
    with torch.no_grad():
        pred, _ = net(batch)
    eng = net.to_engine()
    res = eng.forward([torch.from_numpy(pts).cuda()])
    assert pred[0]["pred_boxes"].shape == res[0]["pred_boxes"].shape
    np.testing.assert_array_equal(pred[0]["pred_labels"].cpu().numpy(), res[0]["pred_labels"].cpu().numpy())
    np.testing.assert_allclose(pred[0]["pred_scores"].cpu().numpy(), res[0]["pred_scores"].cpu().numpy(), atol=1e-5)
    np.testing.assert_allclose(pred[0]["pred_boxes"].cpu().numpy(), res[0]["pred_boxes"].cpu().numpy(), atol=1e-3, rtol=1e-4)
    assert batch["spatial_features"].shape[1] == 256 and batch["encoded_spconv_tensor_stride"] == 8
    assert set(batch["multi_scale_3d_features"]) == {"x_conv1", "x_conv2", "x_conv3", "x_conv4"}


def test_f16x2_range_guard_gives_the_fp32_answer_beyond_fp16_range(hip):
    """VERDICT r2 weak #1 / ADVICE: activations beyond 65504 made the unguarded f16x2 engine return inf / NaN where the
    reference's fp32 convolutions give numbers. With the range guard (default) every conv epilogue records max |out| and the next
    layer pre-scales by a power of two: a model whose first BatchNorm is blown up by 3e4 (activations ~1e5...1e7 all the way
    down) matches the fp32-MFMA engine to 1e-4 of each tensor's magnitude -- sparse levels, BEV map, head maps --, while the
    unguarded engine is loudly non-finite. Both engine and fused module path."""
    from cpd_amd import models
    from cpd_amd import spconv as sp
    from cpd_amd.spconv.pytorch import conv as spc
    cfg = small_cfg()
    sd = init_state_dict(cfg, seed=9)
    sd["backbone_3d.conv_input.1.weight"] = sd["backbone_3d.conv_input.1.weight"] * 3e4
    sd["backbone_3d.conv_input.1.bias"] = sd["backbone_3d.conv_input.1.bias"] * 3e4
    clouds = []
    for s_ in (4, 5, 6, 7):
        p = waymo_cloud(s_, n_points=40000)
        p[:, :2] *= 0.3
        clouds.append(torch.from_numpy(p).cuda())

    reruns = {}

    def run(math, guard=True):
        c = small_cfg()
        c.conv_math, c.range_guard = math, guard
        eng = CenterPointEngine(c, sd, device="cuda")
        out = eng.forward(clouds, return_intermediates=True)[1]
        reruns[(math, guard)] = eng.range_reruns
        return out

    ref, got, bad = run("f32"), run("f16x2"), run("f16x2", guard=False)
    assert reruns == {("f32", True): 0, ("f16x2", True): 1, ("f16x2", False): 0}     # optimistic pass, then ONE guarded re-run

    def close(a, b, what):
        a, b = a.float(), b.float()
        assert torch.isfinite(a).all(), what
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err <= 1e-4, (what, err)

    assert float(ref["levels"]["x_conv2"][0].abs().max()) > 65504          # the case really leaves fp16's range
    for lvl in ("x_conv2", "x_conv3", "x_conv4"):
        assert torch.equal(got["levels"][lvl][1], ref["levels"][lvl][1]) or got["levels"][lvl][1].shape == ref["levels"][lvl][1].shape
    close(got["encoded"][0], ref["encoded"][0], "stride-8 sparse output")
    close(got["bev_cat"], ref["bev_cat"], "BEV concat map")
    close(got["head_rows"][:, :11], ref["head_rows"][:, :11], "head maps")         # (columns 11..15 of the 16-wide block are padding)
    from cpd_amd import ops
    n2 = ref["levels"]["x_conv2"][0].shape[0]
    assert ops.gather_conv_tile(n2, 32, 32, 32, math="f16x2", scaled=True).startswith("rowwave_conv_f16s_kernel")   # the guarded kernels ran
    # unguarded: inf / NaN inside the kernel -- and the ReLU epilogue turns NaN into 0, so the level comes out finite and WRONG
    b2, r2 = bad["levels"]["x_conv2"][0], ref["levels"]["x_conv2"][0]
    assert (not torch.isfinite(b2).all()) or float((b2 - r2).abs().max()) > 1e-2 * float(r2.abs().max())
    # the fused module path carries the blocks as tensor attributes
    mcfg = models.waymo_centerpoint_cfg()
    mcfg.BACKBONE_2D.NUM_FILTERS = cfg.bev_num_filters
    mcfg.BACKBONE_2D.NUM_UPSAMPLE_FILTERS = cfg.bev_num_upsample_filters
    mcfg.BACKBONE_2D.LAYER_NUMS = cfg.bev_layer_nums
    old = spc.default_conv_math()
    try:
        outs = {}
        for math in ("f32", "f16x2"):
            sp.install(conv_math=math)
            net = models.CenterPoint(mcfg, point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size).cuda().eval()
            net.load_state_dict(sd)
            from cpd_amd import ops
            vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, 5, cfg.max_points_per_voxel, cfg.max_voxels)
            _, coords, _, feats, nvox = vox.batch(clouds)
            n = int(nvox[len(clouds)])
            bd = {"voxel_features": feats[:n].clone(), "voxel_coords": coords[:n].float(), "batch_size": len(clouds)}
            with torch.no_grad():
                try:
                    net(bd)
                except Exception:                      # decode of blown-up maps is not what is under test
                    pass
            outs[math] = bd["st_features_2d"]
        close(outs["f16x2"], outs["f32"], "module path BEV map")
    finally:
        spc.set_default_conv_math(old)


@pytest.mark.parametrize("math", ["f32", "f16x2"])
def test_fused_eval_module_path_equals_the_unfused_modules(hip, math):
    """in eval mode without autograd the drop-in modules fuse what the reference writes as separate modules
    (SparseSequential conv + BatchNorm1d + ReLU, SparseBasicBlock, ZeroPad2d + Conv2d + BatchNorm2d + ReLU, the SeparateHead
    branches, batched decode + NMS). Same model, same batch_dict, once fused (torch.no_grad) and once module by module
    (grad mode on: the reference's own sequence of modules): every intermediate the reference exposes agrees to fp32 rounding
    and the detections are the same. Two frames; the package-default arithmetic set through spconv.install(conv_math=...)."""
    from cpd_amd import models, ops
    from cpd_amd import spconv as sp
    from cpd_amd.spconv.pytorch import conv as spc
    cfg = small_cfg()
    cfg.bev_layer_nums = [2, 2]
    mcfg = models.waymo_centerpoint_cfg()
    mcfg.BACKBONE_2D.LAYER_NUMS = [2, 2]
    mcfg.BACKBONE_2D.NUM_FILTERS = cfg.bev_num_filters
    mcfg.BACKBONE_2D.NUM_UPSAMPLE_FILTERS = cfg.bev_num_upsample_filters
    mcfg.DENSE_HEAD.POST_PROCESSING.POST_CENTER_LIMIT_RANGE = cfg.post_center_limit_range
    mcfg.DENSE_HEAD.POST_PROCESSING.MAX_OBJ_PER_SAMPLE = cfg.max_obj_per_sample
    old = spc.default_conv_math()
    sp.install(conv_math=math)
    try:
        net = models.CenterPoint(mcfg, point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size).cuda().eval()
        net.load_state_dict(init_state_dict(cfg, seed=7))
        assert net.backbone_3d.conv2[0][0].conv_math == math and net.backbone_2d.blocks[0][1].math == math
        clouds = []
        for s_ in (2, 3):
            p = waymo_cloud(s_, n_points=40000)
            p[:, :2] *= 0.3
            clouds.append(torch.from_numpy(p).cuda())
        vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, 5, cfg.max_points_per_voxel, cfg.max_voxels)
        _, coords, _, feats, nvox = vox.batch(clouds)
        n = int(nvox[2])

        def batch():
            return {"voxel_features": feats[:n].clone(), "voxel_coords": coords[:n].float(), "batch_size": 2}

        bf, bu = batch(), batch()
        with torch.no_grad():
            pf, _ = net(bf)
        mf = net.dense_head.forward_ret_dict["pred_dicts"][0]
        with torch.enable_grad():
            pu, _ = net(bu)
        tol = dict(atol=2e-4, rtol=1e-4)
        for lvl in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
            a, b = bf["multi_scale_3d_features"][lvl], bu["multi_scale_3d_features"][lvl]
            assert torch.equal(a.indices, b.indices)
            np.testing.assert_allclose(a.features.cpu().numpy(), b.features.detach().cpu().numpy(), **tol)
        assert bf["spatial_features"].shape == bu["spatial_features"].shape
        np.testing.assert_allclose(bf["spatial_features"].cpu().numpy(), bu["spatial_features"].detach().cpu().numpy(), **tol)
        assert bf["spatial_features"].is_contiguous(memory_format=torch.channels_last)
        np.testing.assert_allclose(bf["st_features_2d"].cpu().numpy(), bu["st_features_2d"].detach().cpu().numpy(), **tol)
        assert isinstance(mf, models.HeadMaps)
        with torch.enable_grad():
            mu = net.dense_head.heads_list[0](net.dense_head.shared_conv(bu["st_features_2d"]))
        assert not isinstance(mu, models.HeadMaps)
        for k in mu:
            np.testing.assert_allclose(mf[k].cpu().numpy(), mu[k].detach().cpu().numpy(), **tol)
        for b in range(2):
            assert pf[b]["pred_boxes"].shape == pu[b]["pred_boxes"].shape and pf[b]["pred_boxes"].shape[0] > 0
            np.testing.assert_array_equal(pf[b]["pred_labels"].cpu().numpy(), pu[b]["pred_labels"].cpu().numpy())
            np.testing.assert_allclose(pf[b]["pred_scores"].cpu().numpy(), pu[b]["pred_scores"].detach().cpu().numpy(), atol=2e-5)
            np.testing.assert_allclose(pf[b]["pred_boxes"].cpu().numpy(), pu[b]["pred_boxes"].detach().cpu().numpy(), atol=1e-3, rtol=1e-4)
    finally:
        spc.set_default_conv_math(old)


def test_mm_branch_runs_in_training_mode_only(hip):
    """VoxelResBackBone8x with MM: in training mode `voxel_features1 / voxel_coords1` go through the second encoder
    (spconv_backbone.py:560-598) and come back as multi_scale_3d_features_mm; with the first encoder's weights copied
    into it, its level-1 output equals the main branch's on the same voxels; eval mode does not touch it."""
    from cpd_amd import models
    cfg = models.waymo_centerpoint_cfg()
    cfg.BACKBONE_3D.MM = True
    torch.manual_seed(3)
    bb = models.VoxelResBackBone8x(cfg.BACKBONE_3D, input_channels=5, grid_size=[1504, 1504, 40]).cuda()
    sd = bb.state_dict()
    for k in list(sd):
        for src, dst in (("conv_input.", "conv_input_2."), ("conv1.", "conv1_2.")):
            if k.startswith(src):
                sd[dst + k[len(src):]] = sd[k].clone()
    bb.load_state_dict(sd)
    rng = np.random.default_rng(0)
    zyx = np.unique(rng.integers([0, 700, 700], [41, 800, 800], size=(6000, 3)), axis=0)
    coords = torch.from_numpy(np.pad(zyx, ((0, 0), (1, 0))).astype(np.float32)).cuda()
    feats = torch.randn(coords.shape[0], 5, device="cuda")
    batch = {"voxel_features": feats, "voxel_coords": coords, "voxel_features1": feats.clone(), "voxel_coords1": coords.clone(),
             "batch_size": 1}
    with torch.no_grad():
        out = bb.train()(dict(batch))
        mm, main = out["multi_scale_3d_features_mm"], out["multi_scale_3d_features"]
        assert set(mm) == {"x_conv1", "x_conv2", "x_conv3", "x_conv4"} and out["encoded_spconv_tensor_stride_mm"] == 8
        assert torch.equal(mm["x_conv1"].indices, main["x_conv1"].indices)
        assert torch.allclose(mm["x_conv1"].features, main["x_conv1"].features, atol=1e-5)
        assert mm["x_conv4"].features.shape[1] == 128 and mm["x_conv4"].spatial_shape == main["x_conv4"].spatial_shape
        assert "multi_scale_3d_features_mm" not in bb.eval()(dict(batch))


def test_mm_branch_all_levels_match_oracle(oracle, hip):
    """The `MM: True` prototype encoder (spconv_backbone.py:456-486,560-598: conv_input_2, conv1_2 with two residual blocks,
    conv2_2 / conv3_2 / conv4_2 with a strided conv and ONE residual block each) runs in training mode, i.e. on batch-statistics
    BatchNorm: all four levels of multi_scale_3d_features_mm against the same graph walked on the CPU oracle (its rulebooks and
    sparse convolutions, batch-stat BatchNorm in numpy) -- indices bit-exact in canonical order, features <= 1e-4."""
    from cpd_amd import models
    cfg = models.waymo_centerpoint_cfg()
    cfg.BACKBONE_3D.MM = True
    torch.manual_seed(11)
    grid = [400, 400, 40]
    bb = models.VoxelResBackBone8x(cfg.BACKBONE_3D, input_channels=5, grid_size=grid).cuda().train()
    with torch.no_grad():
        for mod in bb.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.uniform_(0.6, 1.4); mod.bias.normal_(0, 0.2)
    rng = np.random.default_rng(4)
    zyx = np.unique(np.concatenate([rng.integers([0, 100, 100], [41, 260, 260], size=(9000, 3)),
                                    rng.integers([5, 150, 150], [20, 200, 200], size=(6000, 3))]), axis=0)
    rng.shuffle(zyx)
    b = rng.integers(0, 2, (zyx.shape[0], 1))
    coords = np.concatenate([b, zyx], 1).astype(np.int32)
    feats = rng.normal(size=(coords.shape[0], 5)).astype(np.float32)
    batch = {"voxel_features": torch.from_numpy(feats).cuda(), "voxel_coords": torch.from_numpy(coords).float().cuda(),
             "voxel_features1": torch.from_numpy(feats).cuda(), "voxel_coords1": torch.from_numpy(coords).float().cuda(), "batch_size": 2}
    with torch.no_grad():
        mm = bb(batch)["multi_scale_3d_features_mm"]
    sd = {k: v.detach().cpu().numpy() for k, v in bb.state_dict().items()}

    def bn_train(x, name):
        mean = x.astype(np.float64).mean(0)
        var = x.astype(np.float64).var(0)                                  # biased, as BatchNorm normalises in training mode
        y = (x - mean) / np.sqrt(var + 1e-3) * sd[name + ".weight"] + sd[name + ".bias"]
        return y.astype(np.float32)

    relu = lambda v: np.maximum(v, np.float32(0))
    conv = lambda name, x, nbr: oracle.sparse_conv(x, sd[name + ".weight"], sd.get(name + ".bias"), nbr)

    def block(name, x, nbr):
        y = relu(bn_train(conv(name + ".conv1", x, nbr), name + ".bn1"))
        y = bn_train(conv(name + ".conv2", y, nbr), name + ".bn2")
        return relu(y + x)

    shape = [41, 400, 400]
    nbr = oracle.subm_rulebook(coords, 2, shape, [3, 3, 3])
    x = relu(bn_train(conv("conv_input_2.0", feats, nbr), "conv_input_2.1"))
    x = block("conv1_2.0", x, nbr)
    x = block("conv1_2.1", x, nbr)
    want = {"x_conv1": (x, coords)}
    cur = coords
    for i, (stage, k, s, pd) in enumerate([("conv2_2", [3, 3, 3], [2, 2, 2], [1, 1, 1]), ("conv3_2", [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                                           ("conv4_2", [3, 3, 3], [2, 2, 2], [0, 1, 1])], start=2):
        out_idx = oracle.conv_outset(cur, 2, shape, k, s, pd)
        nbr_dn = oracle.conv_rulebook(cur, out_idx, 2, shape, k, s, pd)
        x = relu(bn_train(conv(stage + ".0.0", x, nbr_dn), stage + ".0.1"))
        shape = oracle.conv_out_shape(shape, k, s, pd)
        cur = out_idx
        x = block(stage + ".1", x, oracle.subm_rulebook(cur, 2, shape, [3, 3, 3]))
        want["x_conv%d" % i] = (x, cur)
    for name, (f0, i0) in want.items():
        np.testing.assert_array_equal(mm[name].indices.cpu().numpy(), i0, err_msg=name)
        np.testing.assert_allclose(mm[name].features.cpu().numpy(), f0, atol=1e-4, rtol=0, err_msg=name)


def test_empty_ragged_and_out_of_range_inputs(hip):
    """Edge cases of the batch contract: an empty cloud, a 5-point cloud and a cloud entirely outside the
    range, alone and mixed into a batch; a frame's detections do not depend on its neighbours."""
    cfg = ModelConfig()
    eng = CenterPointEngine(cfg, init_state_dict(cfg, seed=0))
    full = torch.from_numpy(waymo_cloud(0, n_points=40000)).cuda()
    tiny, empty = full[:5].clone(), full[:0].clone()
    outside = full.clone()
    outside[:, 0] += 500.0
    alone = eng.forward([full])[0]
    for batch in ([full, empty, tiny], [outside, full], [tiny], [outside], [empty]):
        res = eng.forward(batch)
        assert len(res) == len(batch)
        for r in res:
            assert torch.isfinite(r["pred_boxes"]).all() and r["pred_boxes"].shape[0] <= cfg.nms_post_maxsize
        for b, pts in enumerate(batch):
            if pts is full:
                assert res[b]["pred_boxes"].shape == alone["pred_boxes"].shape
                d = (res[b]["pred_boxes"][:, None, :] - alone["pred_boxes"][None, :, :]).abs().amax(-1).amin(1)
                assert float(d.max()) < 1e-3
    # no voxels at all: every frame sees the all-zero BEV map, so an empty and an out-of-range cloud agree
    a, b = eng.forward([empty])[0], eng.forward([outside])[0]
    assert torch.equal(a["pred_boxes"], b["pred_boxes"])


def test_engine_internal_row_orders_do_not_change_the_result(hip):
    """Level 0 in canonical vs first-appearance order, strided levels in tap-pattern vs canonical order: the four combinations
    are the same computation over differently ordered rows -- every frame's boxes, scores and labels must agree (row order only
    moves which fp32 partial sums meet in which lane, not the per-row arithmetic: equal to 1e-5)."""
    cfgs = [ModelConfig(voxel_row_order=v, row_order=r, row_order_min_rows=1024) for v in ("canonical", "appearance") for r in ("taps", "canonical")]
    sd = init_state_dict(cfgs[0], seed=0)
    clouds = [torch.from_numpy(waymo_cloud(k, n_points=60000)).cuda() for k in range(3)]
    outs = [CenterPointEngine(c, sd).forward(clouds) for c in cfgs]
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert a["pred_boxes"].shape == b["pred_boxes"].shape
            assert torch.equal(a["pred_labels"], b["pred_labels"])
            assert torch.allclose(a["pred_scores"], b["pred_scores"], atol=1e-5)
            assert torch.allclose(a["pred_boxes"], b["pred_boxes"], atol=1e-4)


def test_fp16_pair_rows_between_sparse_layers_do_not_change_the_result(hip):
    """ModelConfig.pair_rows: levels 2-4 keep their activations as fp16-pair rows between the f16x2 layers (the split made once by
    the producing epilogue). Same partial products as fp32 rows: every exported level (decoded), the BEV map and the detections
    agree with the fp32-row engine; the launch log shows the pair kernels carrying levels 2-4 and none of them when it is off."""
    from cpd_amd import ops
    cfgs = [ModelConfig(pair_rows=p) for p in (True, False)]
    sd = init_state_dict(cfgs[0], seed=0)
    clouds = [torch.from_numpy(waymo_cloud(k, n_points=60000)).cuda() for k in range(3)]
    outs, logs = [], []
    for c in cfgs:
        eng = CenterPointEngine(c, sd)
        with ops.launch_log() as log:
            outs.append(eng.forward(clouds, return_intermediates=True))
        logs.append(log.counts)
        assert eng.range_reruns == 0
    pair_launches = sum(v for k, v in logs[0].items() if k.startswith("rowwave_conv_f16p"))
    assert pair_launches == 15, logs[0]            # 4 + 5 + 5 + conv_out: the layers that read levels 2-4 (conv2.down reads 16 fp32 channels)
    # round 5: at three frames the dense half runs on pair maps too (window / tile pair kernels; the head's output tile writes fp32)
    dense_pair = sum(v for k, v in logs[0].items() if k.startswith(("window_conv_f16p", "tile_conv_f16p")))
    assert dense_pair == 17 and outs[0][1]["dense_pairs"], logs[0]       # 6 + 6 block convs, 2 deblocks, shared conv, the two head launches
    assert not any("f16p" in k for k in logs[1]), logs[1]
    assert not any(k.startswith(("rowwave_conv_f16_kernel", "rowwave_conv_f16e_kernel")) for k in logs[0]), logs[0]
    (ra, ia), (rb, ib) = outs
    for name in ("x_conv2", "x_conv3", "x_conv4"):
        fa, ca, _ = ia["levels"][name]
        fb, cb, _ = ib["levels"][name]
        assert torch.equal(ca, cb)
        assert float((fa - fb).abs().max()) <= 1e-4 * max(1.0, float(fb.abs().max())), name
    assert float((ia["spatial_features_nhwc"] - ib["spatial_features_nhwc"]).abs().max()) <= 1e-4 * max(1.0, float(ib["spatial_features_nhwc"].abs().max()))
    for a, b in zip(ra, rb):
        assert a["pred_boxes"].shape == b["pred_boxes"].shape
        assert torch.equal(a["pred_labels"], b["pred_labels"])
        # (the two engines sum a 32-channel block in different orders: fp32 rounding of the activations, 1e-5 relative, through
        # the dense half and the box decoding)
        assert torch.allclose(a["pred_scores"], b["pred_scores"], atol=1e-4)
        # boxes as SETS (random-init scores tie to 1e-5: two engines may rank equal-score boxes differently): every box of one
        # engine has a box of the other within 2 cm in centre and size
        if a["pred_boxes"].shape[0]:
            near = torch.cdist(a["pred_boxes"][:, :6], b["pred_boxes"][:, :6], p=float("inf")).min(dim=1).values
            assert float((near <= 2e-2).float().mean()) >= 0.99, float(near.max())
    ha, hb = ia["head_rows"][:, :11], ib["head_rows"][:, :11]
    assert float((ha - hb).abs().max()) <= 1e-4 * max(1.0, float(hb.abs().max()))


def test_batch_of_65_frames_does_not_leave_a_group_of_one(hip):
    """more frames than one batched-voxelizer call takes (64) go in groups; 65 used to leave a group of ONE frame, which
    the batched call refuses -- the last group now borrows a frame. Frames 0, 63 and 64 of a 65-frame batch of small clouds give
    the detections they give in a batch of their own."""
    cfg = small_cfg()
    eng = CenterPointEngine(cfg, init_state_dict(cfg, seed=3), device="cuda")
    clouds = []
    for i in range(65):
        p = waymo_cloud(i % 5, n_points=3000 + 17 * i)
        p[:, :2] *= 0.3
        clouds.append(torch.from_numpy(p).cuda())
    res = eng.forward(clouds)
    assert len(res) == 65
    for i in (0, 63, 64):
        one = eng.forward([clouds[i], clouds[(i + 1) % 65]])[0]
        assert res[i]["pred_boxes"].shape == one["pred_boxes"].shape
        np.testing.assert_array_equal(res[i]["pred_labels"].cpu().numpy(), one["pred_labels"].cpu().numpy())
        np.testing.assert_allclose(res[i]["pred_scores"].cpu().numpy(), one["pred_scores"].cpu().numpy(), atol=2e-5)


def test_range_guard_stays_guarded_after_a_rerun(hip):
    """a model that habitually exceeds fp16's range must not pay a second pass on every step: after ONE re-run the engine
    stays on the guarded kernels (RANGE_STICKY_STEPS, extended while activations stay >= 2^14) -- same results, no more re-runs --
    and a model inside the range never enters the guarded mode."""
    cfg = small_cfg()
    sd = init_state_dict(cfg, seed=9)
    hot = dict(sd)
    hot["backbone_3d.conv_input.1.weight"] = sd["backbone_3d.conv_input.1.weight"] * 3e4
    hot["backbone_3d.conv_input.1.bias"] = sd["backbone_3d.conv_input.1.bias"] * 3e4
    clouds = []
    for s_ in (4, 5):
        p = waymo_cloud(s_, n_points=40000)
        p[:, :2] *= 0.3
        clouds.append(torch.from_numpy(p).cuda())
    eng = CenterPointEngine(cfg, hot)
    with pytest.warns(UserWarning, match="range-guarded"):
        _, first = eng.forward(clouds, return_intermediates=True)
    assert (eng.range_reruns, eng.range_guarded_steps) == (1, 1)
    for k in range(3):
        _, again = eng.forward(clouds, return_intermediates=True)
        assert (eng.range_reruns, eng.range_guarded_steps) == (1, 2 + k)        # guarded from the start: no second pass
        assert torch.equal(again["head_rows"][:, :11], first["head_rows"][:, :11])
    assert eng._guard_left == eng.RANGE_STICKY_STEPS                            # still hot: the stay is extended every step
    cool = CenterPointEngine(cfg, sd)
    cool.forward(clouds)
    assert (cool.range_reruns, cool.range_guarded_steps, cool._guard_left) == (0, 0, 0)


def test_eval_container_with_a_training_batchnorm_takes_the_plain_path(hip):
    """BaseBEVBackbone in eval() with one deblock's BatchNorm switched back to .train() used to write nothing
    into its concat buffer for that deblock (uninitialised memory in st_features_2d). The fused path now requires the whole tree
    in eval mode; the result equals the module-by-module forward (batch statistics in that BatchNorm)."""
    from cpd_amd import models
    torch.manual_seed(0)
    mcfg = models.waymo_centerpoint_cfg().BACKBONE_2D
    mcfg.NUM_FILTERS, mcfg.NUM_UPSAMPLE_FILTERS, mcfg.LAYER_NUMS = [32, 64], [64, 64], [1, 1]
    net = models.BaseBEVBackbone(mcfg, input_channels=64).cuda().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.75, 1.25)
    x = torch.randn(2, 64, 24, 24, device="cuda")
    with torch.no_grad():
        fused = net({"spatial_features": x})["st_features_2d"].clone()
    net.deblocks[1][1].train()                               # container stays in eval; this BatchNorm now uses batch statistics
    snap = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        got = net({"spatial_features": x})["st_features_2d"].clone()
    net.load_state_dict(snap)                                # (the training BatchNorm updated its running statistics)
    want = net({"spatial_features": x})["st_features_2d"].detach()      # autograd on: every module runs as the plain reference module
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, want, atol=2e-4, rtol=1e-4)
    assert float((got[:, 64:] - fused[:, 64:]).abs().max()) > 1e-3     # and it differs from the running-statistics answer
    torch.testing.assert_close(got[:, :64], fused[:, :64], atol=2e-4, rtol=1e-4)


def test_voxel_backbone8x_eval_side_by_side_stages_match_oracle(oracle, hip):
    """VoxelBackBone8x in eval mode (spconv_backbone.py:333-393): two stages laid side by side along X on a 4x wide grid, one pass,
    then decompose_tensor per stage -- against the same graph walked on the CPU oracle (folded eval BatchNorm), including the
    reference's open-interval column mask; and in training mode every stage on its own (batch-statistics BatchNorm)."""
    from cpd_amd import models
    cfg = models.waymo_centerpoint_cfg().BACKBONE_3D
    torch.manual_seed(5)
    bb = models.VoxelBackBone8x(cfg, input_channels=5, grid_size=[160, 160, 40]).cuda().eval()
    with torch.no_grad():
        for mod in bb.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.uniform_(0.6, 1.4); mod.bias.normal_(0, 0.2)
                mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.7, 1.3)
    rng = np.random.default_rng(8)
    stages = []
    for s_ in range(2):
        zyx = np.unique(np.concatenate([rng.integers([0, 0, 0], [41, 160, 160], size=(7000, 3)),
                                        rng.integers([5, 40, 40], [20, 100, 100], size=(5000, 3))]), axis=0)
        rng.shuffle(zyx)
        b = rng.integers(0, 2, (zyx.shape[0], 1))
        stages.append((np.concatenate([b, zyx], 1).astype(np.int32), rng.normal(size=(zyx.shape[0], 5)).astype(np.float32)))
    batch = {"batch_size": 2, "transform_param": torch.zeros(2, 2, 8)}
    for i, (c, f) in enumerate(stages):
        sid = "" if i == 0 else str(i)
        batch["voxel_features" + sid] = torch.from_numpy(f).cuda()
        batch["voxel_coords" + sid] = torch.from_numpy(c).float().cuda()
    with torch.no_grad():
        out = bb(dict(batch))
    assert out["multi_scale_3d_features"]["x_conv1"] is None and out["multi_scale_3d_features1"]["x_conv2"] is None
    sd = {k: v.detach().cpu().numpy() for k, v in bb.state_dict().items()}

    def bn(x, name):
        sc = sd[name + ".weight"] / np.sqrt(sd[name + ".running_var"] + 1e-3)
        return np.maximum(x * sc + (sd[name + ".bias"] - sd[name + ".running_mean"] * sc), 0).astype(np.float32)

    conv = lambda name, x, nbr: oracle.sparse_conv(x, sd[name + ".weight"], None, nbr)
    coords = np.concatenate([np.concatenate([c[:, :3], c[:, 3:] + i * 160], 1) for i, (c, _) in enumerate(stages)]).astype(np.int32)
    feats = np.concatenate([f for _, f in stages])
    shape = [41, 160, 640]
    nbr = oracle.subm_rulebook(coords, 2, shape, [3, 3, 3])
    x = bn(conv("conv_input.0", feats, nbr), "conv_input.1")
    x = bn(conv("conv1.0.0", x, nbr), "conv1.0.1")
    cur, levels = coords, {}
    for lvl, (stage, k, s, pd) in enumerate([("conv2", [3, 3, 3], [2, 2, 2], [1, 1, 1]), ("conv3", [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                                             ("conv4", [3, 3, 3], [2, 2, 2], [0, 1, 1])], start=2):
        out_idx = oracle.conv_outset(cur, 2, shape, k, s, pd)
        x = bn(conv(stage + ".0.0", x, oracle.conv_rulebook(cur, out_idx, 2, shape, k, s, pd)), stage + ".0.1")
        shape, cur = oracle.conv_out_shape(shape, k, s, pd), out_idx
        sub = oracle.subm_rulebook(cur, 2, shape, [3, 3, 3])
        x = bn(conv(stage + ".1.0", x, sub), stage + ".1.1")
        x = bn(conv(stage + ".2.0", x, sub), stage + ".2.1")
        levels["x_conv%d" % lvl] = (x, cur, list(shape))
    k, s, pd = [3, 1, 1], [2, 1, 1], [0, 0, 0]
    out_idx = oracle.conv_outset(cur, 2, shape, k, s, pd)
    xo = bn(conv("conv_out.0", x, oracle.conv_rulebook(cur, out_idx, 2, shape, k, s, pd)), "conv_out.1")
    levels["out"] = (xo, out_idx, oracle.conv_out_shape(shape, k, s, pd))

    def check(t, name, i):
        f0, i0, sh = levels[name]
        q = sh[2] // 4
        keep = (i0[:, 3] > i * q) & (i0[:, 3] < (i + 1) * q)              # the reference's open interval (column i * q is dropped)
        want_i = i0[keep].copy(); want_i[:, 3] -= i * q
        assert list(t.spatial_shape) == [sh[0], sh[1], q]
        gi, gf = t.indices.cpu().numpy(), t.features.cpu().numpy()
        o1, o2 = canon(gi), canon(want_i)
        np.testing.assert_array_equal(gi[o1], want_i[o2], err_msg="%s stage %d" % (name, i))
        np.testing.assert_allclose(gf[o1], f0[keep][o2], atol=1e-4, rtol=0, err_msg="%s stage %d" % (name, i))

    for i in range(2):
        sid = "" if i == 0 else str(i)
        check(out["multi_scale_3d_features" + sid]["x_conv3"], "x_conv3", i)
        check(out["multi_scale_3d_features" + sid]["x_conv4"], "x_conv4", i)
        check(out["encoded_spconv_tensor" + sid], "out", i)
        assert out["encoded_spconv_tensor_stride" + sid] == 8
    # training mode: each stage separately, all four levels exported
    bb.train()
    with torch.no_grad():
        tr = bb(dict(batch))
    assert tr["multi_scale_3d_features1"]["x_conv1"].features.shape[0] == stages[1][0].shape[0]
    assert list(tr["encoded_spconv_tensor1"].spatial_shape) == [2, 20, 20]


def test_persistent_dense_map_leaves_no_rows_behind(hip):
    """ModelConfig.persistent_dense_map (ops.DenseMap): the BEV map is scattered into a pre-zeroed persistent buffer and its rows are
    zeroed again after the first BEV conv. Different batches through ONE engine, then the first batch again: bit-identical to its first
    run and to an engine that clears a fresh map every step; a map left dirty (an exception between scatter and clear) is cleared in full."""
    from cpd_amd import ops
    cfg_a, cfg_b = ModelConfig(), ModelConfig(persistent_dense_map=False)
    sd = init_state_dict(cfg_a, seed=0)
    ea, eb = CenterPointEngine(cfg_a, sd), CenterPointEngine(cfg_b, sd)
    batches = [[torch.from_numpy(waymo_cloud(10 * r + k, n_points=50000 + 7000 * k)).cuda() for k in range(2)] for r in range(3)]
    first = ea.forward(batches[0])
    assert any(isinstance(v, ops.DenseMap) for v in ea._bev_cache.values())
    ea.forward(batches[1]); ea.forward(batches[2])
    again = ea.forward(batches[0])
    fresh = eb.forward(batches[0])
    assert not any(isinstance(v, ops.DenseMap) for v in eb._bev_cache.values())
    for x, y, z in zip(first, again, fresh):
        for k in ("pred_boxes", "pred_scores", "pred_labels"):
            assert torch.equal(x[k], y[k]) and torch.equal(x[k], z[k]), k
    dmap = next(v for v in ea._bev_cache.values() if isinstance(v, ops.DenseMap))
    assert float(dmap.buf.abs().max()) == 0.0 and dmap.dirty is None          # between steps the map is all zero
    dmap.buf.fill_(3.0); dmap.dirty = (torch.zeros((1, 4), dtype=torch.int32, device="cuda"), 1)      # as if a step had died half-way
    after = ea.forward(batches[0])
    for x, y in zip(first, after):
        assert torch.equal(x["pred_boxes"], y["pred_boxes"])
    # ONE map for every batch size: a smaller batch runs in the leading frames of the same buffer, a larger one replaces it
    for frames in ([batches[1][0]], batches[0] + [batches[2][1]], [batches[2][0]]):
        got, want = ea.forward(frames), eb.forward(frames)
        for x, y in zip(got, want):
            for k in ("pred_boxes", "pred_scores", "pred_labels"):
                assert torch.equal(x[k], y[k]), (len(frames), k)
        maps = [v for v in ea._bev_cache.values() if isinstance(v, ops.DenseMap)]
        assert len(maps) == 1 and maps[0].batch >= len(frames) and float(maps[0].buf.abs().max()) == 0.0
    assert maps[0].batch == 3


@pytest.mark.parametrize("math", ["f32", "f16x2"])
def test_module_path_in_tap_pattern_row_order_changes_no_result(hip, math):
    """cpd_amd.spconv.install(row_order="taps"): the levels strided SparseConv3d layers produce are kept in the engine's tap-pattern
    row order (chunks of 4096 canonical rows sorted by neighbour pattern). A row's sum does not depend on where the row sits, so: every
    sparse level is the SAME set of (index, feature row) pairs, bit for bit; the BEV map and the detections are identical; and the
    levels really are re-ordered (their index lists differ from the canonical run's). Full-size cloud, two frames: levels 2 and 3 are
    above the 65536-row threshold the order applies from."""
    from cpd_amd import models
    from cpd_amd import spconv as sp
    from cpd_amd.spconv.pytorch import conv as spc
    cfg = ModelConfig()
    sd = init_state_dict(cfg, seed=4)
    vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, cfg.num_point_features, cfg.max_points_per_voxel, cfg.max_voxels)
    clouds = [torch.from_numpy(waymo_cloud(s)).cuda() for s in (0, 1)]
    old_math, old_order = spc.default_conv_math(), spc.default_row_order()
    outs = {}
    try:
        for order in ("canonical", "taps"):
            sp.install(conv_math=math, row_order=order)
            net = models.CenterPoint(point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size).cuda().eval()
            net.load_state_dict(sd)
            with torch.no_grad():
                _, coords, _, feats, nvox = vox.batch(clouds)
                n = int(nvox[2])
                bd = {"voxel_features": feats[:n].clone(), "voxel_coords": coords[:n].clone(), "batch_size": 2}
                preds, _ = net(bd)
            outs[order] = (preds, {k: (t.features.clone(), t.indices.clone()) for k, t in bd["multi_scale_3d_features"].items()},
                           bd["spatial_features"].clone())
    finally:
        spc.set_default_conv_math(old_math); spc.set_default_row_order(old_order)
    (pa, la, sa), (pb, lb, sb) = outs["canonical"], outs["taps"]
    assert sum(len(p["pred_boxes"]) for p in pa) > 10
    for p, q in zip(pa, pb):
        assert all(torch.equal(p[k], q[k]) for k in ("pred_boxes", "pred_scores", "pred_labels"))
    assert torch.equal(sa, sb)
    reordered = 0
    for name in la:
        (fa, ia), (fb, ib) = la[name], lb[name]
        assert ia.shape == ib.shape
        key = lambda i: ((i[:, 0].long() * 64 + i[:, 1]) * 2048 + i[:, 2]) * 2048 + i[:, 3]
        oa, ob = torch.argsort(key(ia)), torch.argsort(key(ib))
        assert torch.equal(ia[oa], ib[ob]) and torch.equal(fa[oa], fb[ob]), name
        reordered += int(not torch.equal(ia, ib))
    assert reordered >= 2, "no level was re-ordered: the comparison would be vacuous"


def test_fast_eval_module_path_matches_the_guarded_modules_and_reruns_out_of_range_steps(hip):
    """cpd_amd.spconv.install(fast_eval=True) (round 6): the fused eval modules run the engine's fast forms -- fp16-pair
    rows between fused sparse layers (kernel names asserted), the unscaled split-fp16 kernels inside an optimistic range pass, the
    level-0 index handed over by the voxelizer (batch_dict["voxel_index"]). Against the same model on the guarded fp32-row path: every
    sparse level the same rows to fp32 rounding (read through `.features`: pair rows decode on demand), the BEV map and the head maps
    to 1e-4, the same detections. Then a batch whose activations leave fp16's range: the step is re-run guarded, the results are the
    guarded path's bit for bit, and the model stays guarded for the sticky steps."""
    import warnings
    from cpd_amd import models
    from cpd_amd import spconv as sp
    from cpd_amd.spconv.pytorch import conv as spc
    cfg = ModelConfig()
    sd = init_state_dict(cfg, seed=4)
    vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, cfg.num_point_features, cfg.max_points_per_voxel, cfg.max_voxels)
    clouds = [torch.from_numpy(waymo_cloud(s)).cuda() for s in (0, 1)]
    old_math, old_order, old_fast = spc.default_conv_math(), spc.default_row_order(), spc.fast_eval()
    outs = {}
    try:
        for fast in (False, True):
            sp.install(conv_math="f16x2", row_order="taps", fast_eval=fast)
            net = models.CenterPoint(point_cloud_range=cfg.point_cloud_range, voxel_size=cfg.voxel_size).cuda().eval()
            net.load_state_dict(sd)
            with torch.no_grad():
                _, coords, _, feats, nvox, index0 = vox.batch(clouds, index_z_extra=1)
                n = int(nvox[2])
                bd = {"voxel_features": feats[:n].clone(), "voxel_coords": coords[:n].clone(), "batch_size": 2}
                if fast:
                    bd["voxel_index"] = index0
                with ops.launch_log() as log:
                    preds, _ = net(bd)
            lv = bd["multi_scale_3d_features"]
            if fast:
                assert all(t._pairs is not None for t in lv.values()), "no level travelled as pair rows"
                assert bd["encoded_spconv_tensor"]._pairs is None                 # conv_out hands HeightCompression fp32 rows
                names = log.counts
                assert any(k.startswith("rowwave_conv_f16pe_kernel") for k in names), names      # pair rows in AND out, LDS epilogue
                assert any(k.startswith("gather_conv_h16_kernel") for k in names), names         # the 16-channel level on pair rows
                assert not any("f16s" in k for k in names), names                                 # nothing ran guarded
                assert net.backbone_3d.conv_input[0].indice_key == "subm1" and bd["voxel_index"] is index0
            outs[fast] = (preds, {k: (t.features.clone(), t.indices.clone()) for k, t in lv.items()}, bd["spatial_features"].clone(),
                          bd["st_features_2d"].clone())
        (pa, la, sa, ca), (pb, lb, sb_, cb) = outs[False], outs[True]
        assert sum(len(p["pred_boxes"]) for p in pa) > 10
        for name in la:
            assert torch.equal(la[name][1], lb[name][1]), name                     # same rows in the same order
            scale = max(1.0, float(la[name][0].abs().max()))
            assert float((la[name][0] - lb[name][0]).abs().max()) <= 1e-4 * scale, name
        assert float((sa - sb_).abs().max()) <= 1e-4 * max(1.0, float(sa.abs().max()))
        assert float((ca - cb).abs().max()) <= 1e-4 * max(1.0, float(ca.abs().max()))
        for p, q in zip(pa, pb):
            assert abs(len(p["pred_boxes"]) - len(q["pred_boxes"])) <= 2
            x, y = p["pred_boxes"].cpu().numpy(), q["pred_boxes"].cpu().numpy()
            # (centres and sizes; the heading is atan2 of two head maps that random-init weights leave near zero: 1e-4 on the maps is
            # 1e-2 rad there, so it is compared on the maps above, not here)
            d = np.abs(x[:, None, :6] - y[None, :, :6]).max(-1).min(1)
            assert (d <= 1e-2).mean() >= 0.98 and np.median(d) <= 1e-3      # (sizes are exp() of maps of scale ~10 that agree to 1e-4 of it)
        # ---- out of range: features x 2^14 push the first layers' activations beyond fp16 -> optimistic pass flags it, guarded re-run
        with torch.no_grad():
            big = {"voxel_features": feats[:n].clone() * 16384.0, "voxel_coords": coords[:n].clone(), "batch_size": 2}
            sp.install(fast_eval=False)
            want, _ = net(dict(big))
            sp.install(fast_eval=True)
            net.range_reruns = 0
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter("always")
                got, _ = net(dict(big))
            assert net.range_reruns == 1 and any("re-run" in str(w.message) for w in wlist)
            # (a ReLU network is positively homogeneous: activations beyond 2^15 mean size logits in the thousands, whose exp() is inf in
            # fp32 on ANY path -- bit-identical to the guarded path is the statement, inf for inf)
            same = lambda a, b: a.shape == b.shape and bool(torch.isclose(a.float(), b.float(), rtol=0, atol=0, equal_nan=True).all())
            for p, q in zip(want, got):
                assert len(q["pred_boxes"]) > 0 and all(same(p[k], q[k]) for k in ("pred_boxes", "pred_scores", "pred_labels"))
            left = net._guard_left
            assert left == net.FAST_EVAL_STICKY_STEPS
            again, _ = net(dict(big))                                              # guarded straight away: no second re-run
            assert net.range_reruns == 1 and net._guard_left == left - 1
            assert all(same(p["pred_boxes"], q["pred_boxes"]) for p, q in zip(want, again))
    finally:
        spc.set_default_conv_math(old_math); spc.set_default_row_order(old_order); spc.set_fast_eval(old_fast)
