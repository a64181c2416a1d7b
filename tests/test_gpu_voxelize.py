"""B1 parity on the GPU: cpd_voxelize vs the serial CPU oracle -- coordinates, order, counts and
payload bit-exact; fused means <= 1e-6."""
import numpy as np
import pytest
import torch

from cpd_amd import ops
from cpd_amd.synthetic import KITTI, KITTI_C1, WAYMO, kitti_cloud, waymo_cloud

pytestmark = pytest.mark.gpu


def run_hip(points, cfg, P=None, maxv=None, coord_cols=3):
    P = P or cfg["max_points_per_voxel"]
    maxv = maxv or cfg["max_voxels"]
    vz = ops.Voxelizer(cfg["voxel_size"], cfg["point_cloud_range"], points.shape[1], P, maxv)
    v, c, n, mean, m = vz(torch.from_numpy(points).cuda(), batch_idx=3, coord_cols=coord_cols)
    return v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy(), mean.cpu().numpy()


def check(oracle, points, cfg, P=None, maxv=None):
    P = P or cfg["max_points_per_voxel"]
    maxv = maxv or cfg["max_voxels"]
    v, c, n, mean = run_hip(points, cfg, P, maxv)
    v0, c0, n0 = oracle.voxelize(points, cfg["voxel_size"], cfg["point_cloud_range"], P, maxv)
    assert c.shape == c0.shape
    np.testing.assert_array_equal(c, c0)
    np.testing.assert_array_equal(n, n0)
    np.testing.assert_array_equal(v, v0)
    np.testing.assert_allclose(mean, oracle.mean_vfe(v0, n0), rtol=1e-6, atol=1e-6)
    return c0.shape[0]


def test_waymo_160k_bit_exact(oracle, hip):
    m = check(oracle, waymo_cloud(0), WAYMO)
    assert m > 50000


def test_kitti_20k_c1_and_c4(oracle, hip):
    pts = kitti_cloud(0)
    check(oracle, pts, KITTI_C1)
    check(oracle, pts, KITTI)


def test_shuffled_points_and_small_caps(oracle, hip):
    rng = np.random.default_rng(5)
    pts = waymo_cloud(1, n_points=40000)
    rng.shuffle(pts)                       # train-mode shuffle (data_processor.py:105-126)
    check(oracle, pts, WAYMO)
    check(oracle, pts, WAYMO, P=2, maxv=1500)      # voxel cap and per-voxel cap both bite
    dense = pts.copy(); dense[:, :2] *= 0.02       # many points per voxel
    check(oracle, dense, WAYMO, P=5, maxv=700)


def test_boundaries_nan_and_ragged(oracle, hip):
    pts = waymo_cloud(2, n_points=5000)
    pts[0, :3] = [-75.2, -75.2, -2.0]; pts[1, :3] = [75.2, 0, 0]; pts[2, :3] = [0, 0, 4.0]
    pts[3, 0] = np.nan; pts[4, 1] = 1e30; pts[5, :3] = [75.19999, 75.19999, 3.99999]; pts[6, 2] = -np.inf
    check(oracle, pts, WAYMO)
    check(oracle, pts[:1], WAYMO)
    # empty cloud
    vz = ops.Voxelizer(WAYMO["voxel_size"], WAYMO["point_cloud_range"], 5, 5, 100)
    v, c, n, mean, m = vz(torch.zeros((0, 5), device="cuda"))
    assert m == 0 and c.shape[0] == 0


def test_batch_column_and_voxel_boundaries(oracle, hip):
    """Points placed exactly on voxel faces: fp32 floor((p-lo)/vs) must round as the CPU does."""
    rng = np.random.default_rng(9)
    k = rng.integers(0, 1504, (20000, 2)).astype(np.float32)
    pts = np.zeros((20000, 5), np.float32)
    pts[:, 0] = np.float32(-75.2) + k[:, 0] * np.float32(0.1)
    pts[:, 1] = np.float32(-75.2) + k[:, 1] * np.float32(0.1)
    pts[:, 2] = np.float32(-2.0) + rng.integers(0, 40, 20000).astype(np.float32) * np.float32(0.15)
    check(oracle, pts, WAYMO)
    v, c, n, mean = run_hip(pts, WAYMO, coord_cols=4)
    assert c.shape[1] == 4 and (c[:, 0] == 3).all()


@pytest.mark.parametrize("max_voxels", [1000000, 900])
def test_batched_voxelizer_equals_per_frame(hip, max_voxels):
    """cpd_voxelize_batch == cpd_voxelize frame by frame (itself bit-exact vs the oracle): same rows, same
    order, same per-frame max_voxels cap, including an empty frame in the middle."""
    import torch
    from cpd_amd import ops
    from cpd_amd.synthetic import WAYMO, waymo_cloud
    vs, rg = WAYMO["voxel_size"], WAYMO["point_cloud_range"]
    frames = [waymo_cloud(0, n_points=30000), waymo_cloud(1, n_points=1000)[:0], waymo_cloud(2, n_points=41000),
              waymo_cloud(3, n_points=5)]
    vox = ops.Voxelizer(vs, rg, 5, 5, max_voxels)
    dev = [torch.from_numpy(f).cuda() for f in frames]
    voxels, coords, num, mean, nvox = vox.batch(dev, want_voxels=True)
    counts = nvox.cpu().numpy()
    row = 0
    for b, f in enumerate(dev):
        v1, c1, n1, m1, k = vox(f, batch_idx=b, coord_cols=4, want_voxels=True, want_mean=True, sync=True)
        assert counts[b] == k
        assert torch.equal(coords[row:row + k], c1) and torch.equal(num[row:row + k], n1)
        assert torch.equal(voxels[row:row + k], v1) and torch.equal(mean[row:row + k], m1)
        row += k
    assert counts[len(frames)] == row


def test_concatenated_buffer_entry_points_equal_the_per_frame_pointer_form(hip):
    """cpd_voxelize_batch / _index / _canonical (ONE concatenated point buffer + offsets) against cpd_voxelize_batch_frames (one device
    pointer per frame: what ops.Voxelizer.batch calls since round 6 -- no torch.cat of the batch's points): identical outputs, an empty
    frame in the middle included, and the site indexes they leave behind answer the same (equal sub-manifold rulebooks)."""
    import ctypes
    import torch
    from cpd_amd import ops
    from cpd_amd._lib import check, farr, iarr, lib, ptr, stream
    from cpd_amd.synthetic import WAYMO, waymo_cloud
    vs, rg = WAYMO["voxel_size"], WAYMO["point_cloud_range"]
    frames = [torch.from_numpy(waymo_cloud(0, n_points=30000)).cuda(), torch.from_numpy(waymo_cloud(1, n_points=10)[:0]).cuda(),
              torch.from_numpy(waymo_cloud(2, n_points=41000)).cuda()]
    vox = ops.Voxelizer(vs, rg, 5, 5, 1000000)
    cat = torch.cat(frames)
    offs = [0, 30000, 30000, 71000]
    n, nf, c = cat.shape[0], 3, 5
    for mode in ("plain", "index", "canonical"):
        if mode == "plain":
            got = vox.batch(frames, want_voxels=True)
        else:
            got = vox.batch(frames, want_voxels=True, index_z_extra=1, canonical=mode == "canonical")
        cap = got[1].shape[0]
        voxels = torch.empty((cap, 5, c), device="cuda"); coords = torch.empty((cap, 4), dtype=torch.int32, device="cuda")
        num = torch.empty((cap,), dtype=torch.int32, device="cuda"); mean = torch.empty((cap, c), device="cuda")
        nvox = torch.zeros((nf + 1,), dtype=torch.int32, device="cuda")
        ws = torch.empty(lib().cpd_voxelize_batch_workspace_bytes(n, nf, 5, 1000000, farr(vs), farr(rg)), dtype=torch.uint8, device="cuda")
        common = (ptr(cat), iarr(offs), nf, c, farr(vs), farr(rg), 5, 1000000, ptr(voxels), ptr(coords), ptr(num), ptr(mean), ptr(nvox), ptr(ws), ws.numel())
        if mode == "plain":
            check(lib().cpd_voxelize_batch(*common, stream()), "cpd_voxelize_batch")
        else:
            g = vox.grid_zyx
            index = ops.SiteIndex(nf, [g[0] + 1, g[1], g[2]], n, cat.device)
            fn = lib().cpd_voxelize_batch_canonical if mode == "canonical" else lib().cpd_voxelize_batch_index
            check(fn(*common, ptr(index.buf), index.buf.numel(), 1, stream()), mode)
            # (the index itself: same builder, same inputs -- its bytes beyond the used words are uninitialised, so it is compared through
            # what it answers: the rulebook built on it)
            k_ = int(nvox[nf])
            assert torch.equal(ops.rulebook_subm(coords[:k_].contiguous(), index), ops.rulebook_subm(got[1][:k_].contiguous(), got[5])), mode
        k = int(nvox[nf])
        assert k == int(got[4][nf]) and k > 1000 and torch.equal(nvox, got[4])
        assert torch.equal(coords[:k], got[1][:k]) and torch.equal(num[:k], got[2][:k]), mode
        assert torch.equal(voxels[:k], got[0][:k]) and torch.equal(mean[:k], got[3][:k]), mode


def test_batched_voxelizer_builds_the_level0_site_index(oracle, hip):
    """cpd_voxelize_batch_index: same voxels as cpd_voxelize_batch, and the site index it leaves behind (over the grid with one
    more z-level, the backbone's sparse_shape) answers every neighbour lookup like an index built from the coordinates
    (cpd_index_build) -- with and without the max_voxels cap dropping voxels (dropped voxels are not sites)."""
    from cpd_amd.synthetic import waymo_cloud
    vs, rg = [0.1, 0.1, 0.15], [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    clouds = [torch.from_numpy(waymo_cloud(s, n_points=40000 + 7000 * s)).cuda() for s in range(3)] + [torch.zeros((0, 5), device="cuda")]
    for max_voxels in (1000000, 9000):
        vox = ops.Voxelizer(vs, rg, 5, 5, max_voxels)
        g = vox.grid_zyx
        shape = [g[0] + 1, g[1], g[2]]
        _, c0, n0, m0, nv0 = vox.batch(clouds)
        _, c1, n1, m1, nv1, index = vox.batch(clouds, index_z_extra=1)
        total = int(nv0[-1])
        assert torch.equal(nv0, nv1) and torch.equal(c0[:total], c1[:total]) and torch.equal(n0[:total], n1[:total])
        assert torch.equal(m0[:total], m1[:total])
        coords = c1[:total].contiguous()
        want = ops.rulebook_subm(coords, ops.SiteIndex.build(coords, len(clouds), shape))
        got = ops.rulebook_subm(coords, index)
        assert torch.equal(got, want) and torch.equal(got.tapmask, want.tapmask)
        np.testing.assert_array_equal(got.cpu().numpy(), oracle.subm_rulebook(coords.cpu().numpy(), len(clouds), shape, [3, 3, 3]))


def test_batched_voxelizer_beyond_2_to_31_cells(hip):
    """30 frames of the Waymo grid = 2.8e9 cells: bitmap positions are formed in 64 bits (frame x cells-per-frame + cell). Every
    frame of the batch must equal the per-frame voxelizer, the last ones (positions above 2^31) in particular, and the index
    built in place must answer lookups like one built from the coordinates."""
    import torch
    from cpd_amd import ops
    from cpd_amd.synthetic import WAYMO, waymo_cloud
    vs, rg = WAYMO["voxel_size"], WAYMO["point_cloud_range"]
    nf = 30
    dev = [torch.from_numpy(waymo_cloud(s % 5, n_points=6000 + 500 * (s % 7))).cuda() for s in range(nf)]
    vox = ops.Voxelizer(vs, rg, 5, 5, 1000000)
    g = vox.grid_zyx
    assert nf * (g[0] + 1) * g[1] * g[2] > (1 << 31) and vox.batch_supported(nf, 1)
    voxels, coords, num, mean, nvox, index = vox.batch(dev, want_voxels=True, index_z_extra=1)
    counts = nvox.cpu().numpy()
    one = ops.Voxelizer(vs, rg, 5, 5, 1000000)
    row = 0
    for b in (0, 11, 23, 24, nf - 1):
        row = int(counts[:b].sum())
        v1, c1, n1, m1, k = one(dev[b], batch_idx=b, coord_cols=4, want_voxels=True, want_mean=True, sync=True)
        assert counts[b] == k
        assert torch.equal(coords[row:row + k], c1) and torch.equal(num[row:row + k], n1)
        assert torch.equal(voxels[row:row + k], v1) and torch.equal(mean[row:row + k], m1)
    total = int(counts[nf])
    cc = coords[:total].contiguous()
    shape = [g[0] + 1, g[1], g[2]]
    want = ops.rulebook_subm(cc, ops.SiteIndex.build(cc, nf, shape))
    got = ops.rulebook_subm(cc, index)
    assert torch.equal(got, want)


def test_batched_voxelizer_canonical_rows(oracle, hip):
    """cpd_voxelize_batch_canonical: the same voxels, point counts and means as the first-appearance form, rows in ascending
    (frame, z, y, x) order, and a canonical site index (row = rank) whose rulebook is the oracle's on the sorted list."""
    import torch
    from cpd_amd import ops
    from cpd_amd.synthetic import waymo_cloud
    vs, rg = [0.1, 0.1, 0.15], [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    clouds = [torch.from_numpy(waymo_cloud(s, n_points=30000 + 9000 * s)).cuda() for s in range(3)]
    clouds += [torch.zeros((0, 5), device="cuda"), torch.from_numpy(waymo_cloud(7, n_points=12000)).cuda()]
    nf = len(clouds)
    vox = ops.Voxelizer(vs, rg, 5, 5, 1000000)
    g = vox.grid_zyx
    shape = [g[0] + 1, g[1], g[2]]
    v0, c0, n0, m0, nv0, _ = vox.batch(clouds, want_voxels=True, index_z_extra=1)
    v1, c1, n1, m1, nv1, index = vox.batch(clouds, want_voxels=True, index_z_extra=1, canonical=True)
    assert torch.equal(nv0, nv1)
    total = int(nv0[-1])
    key = lambda c: ((c[:, 0].long() * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]
    k1 = key(c1[:total])
    assert bool((k1[1:] > k1[:-1]).all())                                        # strictly ascending: canonical, no duplicates
    order = torch.argsort(key(c0[:total]))
    assert torch.equal(c0[:total][order], c1[:total]) and torch.equal(n0[:total][order], n1[:total])
    assert torch.equal(m0[:total][order], m1[:total]) and torch.equal(v0[:total][order], v1[:total])
    cc = c1[:total].contiguous()
    got = ops.rulebook_subm(cc, index)
    np.testing.assert_array_equal(got.cpu().numpy(), oracle.subm_rulebook(cc.cpu().numpy(), nf, shape, [3, 3, 3]))
    small = ops.Voxelizer(vs, rg, 5, 5, 900)                                    # row capacity limited by the cap: refused, not wrong
    with pytest.raises(Exception):
        small.batch(clouds, index_z_extra=1, canonical=True)


@pytest.mark.gpu
@pytest.mark.parametrize("max_points", [1, 5, 35])
def test_canonical_rows_keep_the_smallest_indices_of_crowded_voxels(oracle, hip, max_points):
    """The canonical form builds its per-voxel point lists by counting sort (arrival slots + a scan) and then keeps the max_points
    SMALLEST point indices of a voxel in ascending order. Coarse voxels (hundreds of points each), shuffled points (a voxel's points
    arrive from all over the cloud, in any order), max_points below / at / above typical counts: rows, kept points, counts and means
    equal the oracle's (serial first-appearance voxelizer) on the sorted voxel list, bit for bit."""
    import torch
    from cpd_amd import ops
    from cpd_amd.synthetic import waymo_cloud
    vs, rg = [0.8, 0.8, 1.0], [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
    rng = np.random.default_rng(max_points)
    clouds_np = []
    for s in range(3):
        p = waymo_cloud(s, n_points=40000 + 7000 * s)
        clouds_np.append(p[rng.permutation(len(p))])
    clouds = [torch.from_numpy(p).cuda() for p in clouds_np]
    vox = ops.Voxelizer(vs, rg, 5, max_points, 200000)
    g = vox.grid_zyx
    shape = [g[0] + 1, g[1], g[2]]
    v1, c1, n1, m1, nv1, _ = vox.batch(clouds, want_voxels=True, index_z_extra=1, canonical=True)
    row = 0
    for f, p in enumerate(clouds_np):
        vo, co, no = oracle.voxelize(p, vs, rg, max_points, 200000)
        mo = oracle.mean_vfe(vo, no)
        k = (co[:, 0].astype(np.int64) * shape[1] + co[:, 1]) * shape[2] + co[:, 2]
        o = np.argsort(k)
        m = len(o)
        assert int(nv1[f]) == m
        assert int(no.max()) == max_points or max_points == 35
        np.testing.assert_array_equal(c1[row:row + m].cpu().numpy(), np.concatenate([np.full((m, 1), f, np.int32), co[o]], 1))
        np.testing.assert_array_equal(n1[row:row + m].cpu().numpy(), no[o])
        np.testing.assert_array_equal(v1[row:row + m].cpu().numpy(), vo[o])
        np.testing.assert_allclose(m1[row:row + m].cpu().numpy(), mo[o], rtol=1e-6, atol=1e-6)
        row += m
    assert int(nv1[-1]) == row

