"""RoI-head feature pooling: HIP kernels vs the oracle (bit-exact indices) and the fused
pooling module vs the output of the reference's own Python module (fixture roi_pool.npz, <= 1e-4)."""
import numpy as np
import pytest
import torch

from cpd_amd import ops

pytestmark = pytest.mark.gpu


def _scene(rng, batch, shape, n_vox, n_query):
    cells = np.unique(np.stack([rng.integers(0, batch, n_vox)] + [rng.integers(0, s, n_vox) for s in shape], 1), axis=0).astype(np.int32)
    xyz = ((cells[:, [3, 2, 1]] + 0.5) * np.array([0.4, 0.4, 0.6]) + np.array([-3.0, -2.0, -1.0])).astype(np.float32)
    q = np.stack([rng.integers(0, batch, n_query)] + [rng.integers(-1, s + 1, n_query) for s in shape], 1).astype(np.int32)
    q[:, 1:] = np.clip(q[:, 1:], 0, np.array(shape) - 1)
    q = q[np.argsort(q[:, 0], kind="stable")]
    qxyz = ((q[:, [3, 2, 1]] + rng.uniform(0, 1, (n_query, 3))) * np.array([0.4, 0.4, 0.6]) + np.array([-3.0, -2.0, -1.0])).astype(np.float32)
    return cells, np.ascontiguousarray(xyz), np.ascontiguousarray(q), np.ascontiguousarray(qxyz)


@pytest.mark.parametrize("max_range,radius,nsample", [([1, 1, 1], 0.7, 4), ([2, 2, 2], 1.1, 16), ([1, 4, 4], 1.5, 16),
                                                      ([4, 4, 4], 1.6, 16), ([8, 8, 8], 3.2, 16), ([8, 8, 8], 0.5, 16),   # the shipped yaml's ranges
                                                      ([1, 2, 15], 2.5, 16), ([1, 1, 16], 2.5, 16)])    # 31-cell rows; 33: the cell-wise kernel
def test_voxel_query_dense_and_indexed_match_oracle(oracle, hip, max_range, radius, nsample):
    from cpd_amd import ops, roi_pool
    rng = np.random.default_rng(sum(max_range) + nsample)
    batch, shape = 3, [7, 30, 34]
    cells, xyz, q, qxyz = _scene(rng, batch, shape, 6000, 5003)
    v2p = oracle.voxel2pinds(cells, batch, shape)
    want = oracle.voxel_query(max_range, radius, nsample, xyz, qxyz, q, v2p)
    d_cells, d_xyz, d_q, d_qxyz = (torch.from_numpy(a).cuda() for a in (cells, xyz, q, qxyz))
    got_v2p = roi_pool.generate_voxel2pinds(d_cells, batch, shape)
    np.testing.assert_array_equal(got_v2p.cpu().numpy(), v2p)
    empty_want = want[:, 0] == -1
    want[empty_want] = 0
    idx, empty = roi_pool.voxel_query(max_range, radius, nsample, d_xyz, d_qxyz, d_q, point_indices=got_v2p)
    np.testing.assert_array_equal(empty.cpu().numpy(), empty_want)
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    index = ops.SiteIndex.build(d_cells, batch, shape)           # cells are in canonical order: rank == row
    with ops.launch_log() as log:
        idx2, empty2 = roi_pool.voxel_query(max_range, radius, nsample, d_xyz, d_qxyz, d_q, index=index)
    # the bitmap index is scanned a window ROW (<= 32 bits of the bitmap) at a time; wider rows take the cell-by-cell kernel
    assert log.counts == {"voxel_query_rows_kernel" if 2 * max_range[2] + 1 <= 32 else "voxel_query_kernel<IndexLookup>": 1}, log.counts
    np.testing.assert_array_equal(idx2.cpu().numpy(), want)
    # arbitrary row order goes through the index permutation
    perm = rng.permutation(cells.shape[0])
    index_p = ops.SiteIndex.build(torch.from_numpy(cells[perm]).cuda(), batch, shape)
    idx3, _ = roi_pool.voxel_query(max_range, radius, nsample, torch.from_numpy(xyz[perm]).cuda(), d_qxyz, d_q, index=index_p)
    v2p_p = oracle.voxel2pinds(cells[perm], batch, shape)
    want_p = oracle.voxel_query(max_range, radius, nsample, xyz[perm], qxyz, q, v2p_p)
    want_p[want_p[:, 0] == -1] = 0
    np.testing.assert_array_equal(idx3.cpu().numpy(), want_p)


@pytest.mark.parametrize("max_range,radius,nsample", [([2, 2, 2], 0.4, 16), ([4, 4, 4], 0.8, 16), ([2, 2, 2], 0.8, 16), ([4, 4, 4], 1.6, 16),
                                                      ([8, 8, 8], 1.6, 16), ([1, 2, 15], 2.5, 8), ([3, 3, 3], 0.0, 4)])
def test_voxel_query_from_cell_geometry_matches_oracle(oracle, hip, max_range, radius, nsample):
    """cpd_voxel_query_index_grid: xyz = get_voxel_centers(level) is not loaded but evaluated from the cell coordinates (the
    reference's three fp32 operations per axis), rows of the window beyond the radius are skipped -- same neighbours, same order as
    the oracle's cell-by-cell scan over the xyz rows (voxel_query_gpu.cu:41-77). Query points: random, on the grid's rim, and exactly
    ON cell centres (distances that equal the radius exactly: the test is `not >`)."""
    from cpd_amd import ops, roi_pool
    rng = np.random.default_rng(sum(max_range) + nsample + int(10 * radius))
    batch, shape, stride = 2, [9, 40, 70], 4
    voxel_size, pc_range = [0.1, 0.1, 0.15], [-3.0, -2.0, -1.0, 25.0, 14.0, 4.4]
    cells = np.unique(np.stack([rng.integers(0, batch, 20000)] + [rng.integers(0, s, 20000) for s in shape], 1), axis=0).astype(np.int32)
    d_cells = torch.from_numpy(cells).cuda()
    d_xyz = roi_pool.get_voxel_centers(d_cells[:, 1:4], stride, voxel_size, pc_range).contiguous()
    xyz = d_xyz.cpu().numpy()
    n_query = 6001
    q = np.stack([rng.integers(0, batch, n_query)] + [rng.integers(0, s, n_query) for s in shape], 1).astype(np.int32)
    q = q[np.argsort(q[:, 0], kind="stable")]
    cell = (np.float32(voxel_size) * np.float32(stride)).astype(np.float32)
    qxyz = ((q[:, [3, 2, 1]] + rng.uniform(0, 1, (n_query, 3))) * cell + np.float32(pc_range[:3])).astype(np.float32)
    on_centre = rng.random(n_query) < 0.3
    qxyz[on_centre] = ((q[on_centre][:, [3, 2, 1]].astype(np.float32) + np.float32(0.5)) * cell + np.float32(pc_range[:3])).astype(np.float32)
    qxyz, q = np.ascontiguousarray(qxyz), np.ascontiguousarray(q)
    v2p = oracle.voxel2pinds(cells, batch, shape)
    want = oracle.voxel_query(max_range, radius, nsample, xyz, qxyz, q, v2p)
    empty_want = want[:, 0] == -1
    want[empty_want] = 0
    assert empty_want.sum() < n_query or radius == 0.0
    index = ops.SiteIndex.build(d_cells, batch, shape)
    grid = roi_pool.cell_geometry(voxel_size, stride, pc_range)
    with ops.launch_log() as log:
        idx, empty = roi_pool.voxel_query(max_range, radius, nsample, d_xyz, torch.from_numpy(qxyz).cuda(), torch.from_numpy(q).cuda(),
                                          index=index, grid=grid)
    assert log.counts == {"voxel_query_grid_kernel": 1}, log.counts
    np.testing.assert_array_equal(empty.cpu().numpy(), empty_want)
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    # rows in arbitrary order: the hits go through the index's rank -> row map
    perm = rng.permutation(cells.shape[0])
    index_p = ops.SiteIndex.build(torch.from_numpy(cells[perm]).cuda(), batch, shape)
    idx_p, _ = roi_pool.voxel_query(max_range, radius, nsample, d_xyz[torch.from_numpy(perm).cuda()].contiguous(), torch.from_numpy(qxyz).cuda(),
                                    torch.from_numpy(q).cuda(), index=index_p, grid=grid)
    want_p = oracle.voxel_query(max_range, radius, nsample, xyz[perm], qxyz, q, oracle.voxel2pinds(cells[perm], batch, shape))
    want_p[want_p[:, 0] == -1] = 0
    np.testing.assert_array_equal(idx_p.cpu().numpy(), want_p)


@pytest.mark.parametrize("c,c2", [(16, 24), (32, 32), (64, 48), (32, 64)])
def test_pool_max_mlp_equals_pool_max_followed_by_the_gemm(hip, c, c2):
    """cpd_voxel_pool_max_mlp (pooling + the module's output MLP in one kernel, written into a column block of wider rows) against
    cpd_voxel_pool_max followed by the 1 x 1 GEMM it replaces (voxel_pool_modules.py:96-121), incl. empty balls and a last block of
    points that is not full."""
    from cpd_amd import ops, roi_pool
    from cpd_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(c + c2)
    n, m, ns = 5000, 4099, 16
    fin = torch.randn(n, c, generator=g).cuda()
    xyz = (torch.rand(n, 3, generator=g) * 10).cuda()
    new_xyz = (torch.rand(m, 3, generator=g) * 10).cuda()
    idx = torch.randint(0, n, (m, ns), generator=g, dtype=torch.int32).cuda()
    idx[::7, 0] = -1                                                        # empty balls
    w_pos, b_pos = torch.randn(3, c, generator=g).cuda(), torch.randn(c, generator=g).cuda()
    w_out, t_out = (torch.randn(c, c2, generator=g) * 0.2).cuda(), torch.randn(c2, generator=g).cuda()
    pooled = roi_pool.voxel_pool_max(fin, xyz, new_xyz, idx, w_pos, b_pos)
    want = torch.relu(pooled @ w_out + t_out)
    wide = torch.full((m, c2 + 40), -7.0, device="cuda")
    with ops.launch_log() as log:
        check(lib().cpd_voxel_pool_max_mlp(m, c, ns, ptr(fin), fin.stride(0), ptr(xyz), ptr(new_xyz), ptr(idx), ptr(w_pos), ptr(b_pos), ptr(w_out),
                                           ptr(t_out), c2, 1, wide[:, 24:].data_ptr(), wide.stride(0), stream()), "cpd_voxel_pool_max_mlp")
    assert log.counts == {"voxel_pool_max_mlp_kernel": 1}
    np.testing.assert_allclose(wide[:, 24:24 + c2].cpu().numpy(), want.cpu().numpy(), atol=2e-5, rtol=1e-5)
    assert bool((wide[:, :24] == -7.0).all()) and bool((wide[:, 24 + c2:] == -7.0).all())      # nothing outside the block is touched
    # unsupported shapes are refused, not approximated
    assert lib().cpd_voxel_pool_max_mlp(m, 48, ns, ptr(fin), 48, ptr(xyz), ptr(new_xyz), ptr(idx), ptr(w_pos), ptr(b_pos), ptr(w_out), ptr(t_out),
                                        c2, 1, wide.data_ptr(), wide.stride(0), stream()) != 0


def test_group_points_matches_oracle(oracle, hip):
    from cpd_amd import roi_pool
    rng = np.random.default_rng(5)
    feat = rng.normal(size=(900, 24)).astype(np.float32)
    fcnt, icnt = np.array([400, 500], np.int32), np.array([130, 71], np.int32)
    idx = np.concatenate([rng.integers(0, 400, (130, 16)), rng.integers(0, 500, (71, 16))]).astype(np.int32)
    want = oracle.group_points(feat, fcnt, idx, icnt)
    got = roi_pool.grouping_operation(torch.from_numpy(feat).cuda(), torch.from_numpy(fcnt).cuda(), torch.from_numpy(idx).cuda(),
                                      torch.from_numpy(icnt).cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_pool_module_reproduces_reference_module_output(golden, hip):
    """NeighborVoxelSAModuleMSG with the reference module's own state_dict, on the reference's inputs."""
    from cpd_amd import ops, roi_pool
    g = golden("roi_pool")
    mod = roi_pool.NeighborVoxelSAModuleMSG(query_ranges=[[1, 2, 2], [2, 4, 4]], radii=[0.9, 1.7], nsamples=[8, 16],
                                            mlps=[[16, 16, 24], [16, 16, 24]])
    sd = {k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd.")}
    mod.load_state_dict(sd)
    mod = mod.cuda().eval()
    cells = torch.from_numpy(g["cells"]).cuda()
    shape = [int(v) for v in g["shape"]]
    xyz = roi_pool.get_voxel_centers(cells[:, 1:4], int(g["stride"]), g["voxel_size"].tolist(), g["pcr"].tolist())
    np.testing.assert_allclose(xyz.cpu().numpy(), g["xyz"], atol=1e-6)
    rois = torch.from_numpy(g["rois"]).cuda()
    grid, _ = roi_pool.get_global_grid_points_of_roi(rois, int(g["grid_size"]))
    np.testing.assert_allclose(grid.view(2, -1, 3).cpu().numpy(), g["grid_xyz"], atol=1e-5)
    new_xyz = torch.from_numpy(g["grid_xyz"]).cuda().view(-1, 3).contiguous()
    nc = torch.from_numpy(g["new_coords_bxyz"]).cuda()
    cnt = torch.bincount(cells[:, 0].long(), minlength=2).int()
    new_cnt = torch.full((2,), new_xyz.shape[0] // 2, dtype=torch.int32, device="cuda")
    feats = torch.from_numpy(g["feats"]).cuda()
    v2p = roi_pool.generate_voxel2pinds(cells, 2, shape)
    out = mod(xyz.contiguous(), cnt, new_xyz, new_cnt, nc, feats, voxel2point_indices=v2p)
    np.testing.assert_allclose(out.cpu().numpy(), g["pooled"], atol=1e-4, rtol=0)
    index = ops.SiteIndex.build(cells, 2, shape)
    out2 = mod(xyz.contiguous(), cnt, new_xyz, new_cnt, nc, feats, index=index)
    assert torch.equal(out, out2)


def test_roi_grid_pool_on_engine_levels(hip):
    """End to end on the hot path's own multi-scale features: detections -> RoI grid -> pooled (B*N, 216, 128)."""
    from cpd_amd import ops, roi_pool
    from cpd_amd.engine import CenterPointEngine, ModelConfig, init_state_dict
    from cpd_amd.synthetic import waymo_cloud
    cfg = ModelConfig()
    eng = CenterPointEngine(cfg, init_state_dict(cfg, seed=0))
    pts = [torch.from_numpy(waymo_cloud(i)).cuda() for i in range(2)]
    res, it = eng.forward(pts, return_intermediates=True)
    n_roi = 64
    rois = torch.zeros((2, n_roi, 7), device="cuda")
    for b in range(2):
        k = min(n_roi, res[b]["pred_boxes"].shape[0])
        rois[b, :k] = res[b]["pred_boxes"][:k]
        rois[b, k:, 3:6] = 1.0
    torch.manual_seed(0)
    layers = {name: roi_pool.NeighborVoxelSAModuleMSG(query_ranges=[[2, 2, 2], [4, 4, 4]], radii=r, nsamples=[16, 16],
                                                      mlps=[[c, 32, 32], [c, 32, 32]]).cuda().eval()
              for name, r, c in (("x_conv3", [0.4, 0.8], 64), ("x_conv4", [0.8, 1.6], 128))}
    strides = {"x_conv3": 4, "x_conv4": 8}
    dense = roi_pool.roi_grid_pool(rois, it["levels"], strides, layers, 6, cfg.voxel_size, cfg.point_cloud_range, 2)
    assert dense.shape == (2 * n_roi, 216, 128) and torch.isfinite(dense).all()
    indexes = {}
    for name in layers:
        f, c, s = it["levels"][name]
        indexes[name] = ops.SiteIndex.build(c, 2, s)
    via_index = roi_pool.roi_grid_pool(rois, it["levels"], strides, layers, 6, cfg.voxel_size, cfg.point_cloud_range, 2, indexes=indexes)
    assert torch.equal(dense, via_index)


def test_proposal_layer_matches_reference_nms_golden(golden, hip):
    """proposal_layer = per-sample max class score + class_agnostic_nms; checked on the reference's own
    class_agnostic_nms fixture (tests/golden/nms.npz) replicated into a batch with different class layouts."""
    from cpd_amd import roi_pool
    g = golden("nms")
    boxes, scores, want = g["n500_t3.boxes"], g["n500_t3.scores"], g["n500_t3.selected"]
    thr, pre, post = float(g["n500_t3.thr"]), 4096, 500                # make_golden.py: NMS_PRE_MAXSIZE 4096, NMS_POST_MAXSIZE 500
    n = boxes.shape[0]
    cls = np.full((2, n, 3), -5.0, np.float32)
    cls[0, :, 1] = scores                                          # sample 0: everything is class 2
    cls[1, np.arange(n), np.arange(n) % 3] = scores                # sample 1: classes cycle
    bb = np.stack([boxes, boxes]).astype(np.float32)
    rois, rs, rl, kept = roi_pool.proposal_layer(torch.from_numpy(bb).cuda(), torch.from_numpy(cls).cuda(), thr, pre, post)
    k = len(want)
    assert kept.tolist() == [k, k]
    for b in range(2):
        np.testing.assert_array_equal(rois[b, :k].cpu().numpy(), boxes[want])
        np.testing.assert_array_equal(rs[b, :k].cpu().numpy(), scores[want])
        assert float(rois[b, k:].abs().sum()) == 0.0
    assert (rl[0, :k] == 2).all()
    np.testing.assert_array_equal(rl[1, :k].cpu().numpy(), want % 3 + 1)


def test_voxel_rcnn_head_eval_forward_matches_reference_module(golden, hip):
    """Second stage end to end: the reference's own VoxelRCNNHead output (run on the oracle-backed extension) for the
    same state_dict, RoIs and multi-scale features."""
    from cpd_amd import roi_pool
    g = golden("voxel_rcnn_head")
    cfg = dict(ROI_GRID_POOL=dict(FEATURES_SOURCE=["x_conv3", "x_conv4"], GRID_SIZE=3, POOL_LAYERS=dict(
        x_conv3=dict(MLPS=[[16, 16], [16, 16]], QUERY_RANGES=[[1, 1, 1], [2, 2, 2]], POOL_RADIUS=[0.6, 1.2], NSAMPLE=[8, 8], POOL_METHOD="max_pool"),
        x_conv4=dict(MLPS=[[16, 16], [16, 16]], QUERY_RANGES=[[1, 1, 1], [2, 2, 2]], POOL_RADIUS=[1.2, 2.4], NSAMPLE=[8, 8], POOL_METHOD="max_pool"))),
        SHARED_FC=[64, 64], CLS_FC=[32], REG_FC=[32], DP_RATIO=0.3)
    head = roi_pool.VoxelRCNNHead({"x_conv3": 24, "x_conv4": 32}, cfg, point_cloud_range=g["pcr"].tolist(), voxel_size=[0.1, 0.1, 0.15], num_class=1)
    sd = {k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("rh.")}
    res = head.load_state_dict(sd, strict=False)
    assert not res.missing_keys, res.missing_keys
    head = head.cuda().eval()
    lv = {"x_conv3": (torch.from_numpy(g["c3_feat"]).cuda(), torch.from_numpy(g["c3_idx"]).cuda(), [11, 104, 104]),
          "x_conv4": (torch.from_numpy(g["c4_feat"]).cuda(), torch.from_numpy(g["c4_idx"]).cuda(), [5, 52, 52])}
    bd = {"batch_size": 2, "rois": torch.from_numpy(g["rois"]).cuda(), "multi_scale_3d_features": lv,
          "multi_scale_3d_strides": {"x_conv3": 4, "x_conv4": 8}}
    out = head(bd)
    np.testing.assert_allclose(out["batch_cls_preds"].cpu().numpy(), g["batch_cls_preds"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out["batch_box_preds"].cpu().numpy(), g["batch_box_preds"], atol=2e-4, rtol=1e-5)


def test_roi_grid_points_kernel_matches_the_torch_sequence(hip):
    """cpd_roi_grid_points (one launch) against the reference's torch sequence it replaces -- get_global_grid_points_of_roi, the three
    float floor divisions, cat, int (voxel_rcnn_head.py:186-273, 365-386): grid points to fp32 rounding (the device cos / sin of torch's
    build and of this library's may differ in the last bit), cells equal except where a point sits within that rounding of a cell face."""
    from cpd_amd import roi_pool
    g = torch.Generator().manual_seed(4)
    B, N, G = 3, 257, 6
    rois = torch.cat([torch.rand(B, N, 2, generator=g) * 150.4 - 75.2, torch.rand(B, N, 1, generator=g) * 6 - 2, torch.rand(B, N, 3, generator=g) * 6 + 0.3,
                      torch.rand(B, N, 1, generator=g) * 7 - 3.5], dim=-1).cuda()
    rois[1, 100:] = 0                                                          # padded (zero) RoIs
    vs, pcr = [0.1, 0.1, 0.15], [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    with ops.launch_log() as log:
        grid, cells = roi_pool.roi_grid_points(rois, G, vs, pcr, strides=[4, 8])
    assert log.counts == {"roi_grid_points_kernel": 1}
    want, _ = roi_pool.get_global_grid_points_of_roi(rois.clone(), G)
    want = want.view(B, -1, 3)
    np.testing.assert_allclose(grid.view(B, -1, 3).cpu().numpy(), want.cpu().numpy(), atol=2e-5, rtol=0)
    gc = torch.cat([(want[:, :, 0:1] - pcr[0]) // vs[0], (want[:, :, 1:2] - pcr[1]) // vs[1], (want[:, :, 2:3] - pcr[2]) // vs[2]], dim=-1)
    bidx = torch.arange(B, device="cuda", dtype=gc.dtype).view(-1, 1, 1).expand(-1, gc.shape[1], 1)
    for stride in (4, 8):
        cur = torch.cat([bidx, gc // stride], dim=-1).int().view(-1, 4)
        same = (cells[stride] == cur).all(dim=1)
        assert float(same.float().mean()) >= 0.9995, float(same.float().mean())
        assert int((cells[stride] - cur).abs().max()) <= 1                    # a differing cell is the neighbour across the face
    # on ITS OWN grid points the kernel's cells are exactly torch's floor divisions (the arithmetic, separated from the cos / sin bits)
    gc2 = torch.stack([(grid[:, 0] - pcr[0]) // vs[0], (grid[:, 1] - pcr[1]) // vs[1], (grid[:, 2] - pcr[2]) // vs[2]], dim=-1)
    b2 = torch.arange(B, device="cuda").repeat_interleave(N * G ** 3).view(-1, 1).to(gc2.dtype)
    for stride in (4, 8):
        assert torch.equal(cells[stride], torch.cat([b2, gc2 // stride], dim=-1).int())
    _, zyx = roi_pool.roi_grid_points(rois, G, vs, pcr, strides=[4], bzyx=True)
    assert torch.equal(zyx[4], cells[4][:, [0, 3, 2, 1]])
