"""BASELINE.json configs as named test cases.
  C1 (CPU, plumbing): 20k-point KITTI-shape cloud, 0.1 m voxels, 3 x SubMConv3d(+BN+ReLU) on the CPU
     oracle, checked against the dense conv3d definition.
  C4 / C5 live in the GPU files (test_gpu_sparse.py::test_kitti_c4_subm_path, below for C5)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cpd_amd.synthetic import KITTI_C1, WAYMO, kitti_cloud, waymo_cloud


def test_c1_cpu_plumbing(oracle):
    cfg = KITTI_C1
    pts = kitti_cloud(0)
    v, c, n = oracle.voxelize(pts, cfg["voxel_size"], cfg["point_cloud_range"], 5, cfg["max_voxels"])
    assert oracle.grid_size(cfg["voxel_size"], cfg["point_cloud_range"]) == [40, 800, 704]
    feat = oracle.mean_vfe(v, n)
    idx = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    shape = [41, 800, 704]
    nbr = oracle.subm_rulebook(idx, 1, shape, [3, 3, 3])
    rng = np.random.default_rng(0)
    x = feat
    chans = [4, 16, 16, 16]
    # crop for the dense check: sites inside a small window (dense conv3d of the full grid is too big)
    win = (idx[:, 2] >= 300) & (idx[:, 2] < 364) & (idx[:, 3] >= 100) & (idx[:, 3] < 164)
    dense = None
    for l in range(3):
        w = (rng.normal(size=(chans[l + 1], 3, 3, 3, chans[l])) * np.sqrt(2.0 / (27 * chans[l]))).astype(np.float32)
        scale = rng.uniform(0.5, 1.5, chans[l + 1]).astype(np.float32); shift = rng.normal(size=chans[l + 1]).astype(np.float32) * 0.1
        y = oracle.affine_rows(oracle.sparse_conv(x, w, None, nbr), scale, shift, None, True)
        if l == 0:   # definition check on the first layer (later layers depend on sites outside the crop)
            sub = idx[win]
            d = np.zeros((1, chans[0], 41, 66, 66), np.float32)
            inner = (idx[:, 2] >= 299) & (idx[:, 2] < 365) & (idx[:, 3] >= 99) & (idx[:, 3] < 165)
            ii = idx[inner]
            d[0, :, ii[:, 1], ii[:, 2] - 299, ii[:, 3] - 99] = x[inner]
            ref = F.conv3d(torch.from_numpy(d), torch.from_numpy(w).permute(0, 4, 1, 2, 3).contiguous(), padding=1).numpy()
            want = np.maximum(ref[0, :, sub[:, 1], sub[:, 2] - 299, sub[:, 3] - 99] * scale + shift, 0)
            np.testing.assert_allclose(y[win], want, atol=1e-4, rtol=0)
        x = y
    assert x.shape == (idx.shape[0], 16) and np.isfinite(x).all() and (x >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n_points", [160000, 1000000])
def test_c5_dense_stress_rulebook(oracle, hip, n_points):
    """C5: Waymo-shape cloud at (0.05, 0.05, 0.1) m voxels -> sparse [61, 3008, 3008]; voxelizer,
    site index and SubM rulebook bit-exact against the oracle (160k and the 1M-point variant)."""
    from cpd_amd import ops
    cfg = dict(WAYMO, voxel_size=[0.05, 0.05, 0.1])
    pts = waymo_cloud(3, n_points=n_points, n_az=2650 if n_points <= 160000 else 18000)
    vz = ops.Voxelizer(cfg["voxel_size"], cfg["point_cloud_range"], 5, 5, cfg["max_voxels"])
    _, c, n, mean, m = vz(torch.from_numpy(pts).cuda(), batch_idx=0, coord_cols=4, want_voxels=False)
    v0, c0, n0 = oracle.voxelize(pts, cfg["voxel_size"], cfg["point_cloud_range"], 5, cfg["max_voxels"])
    np.testing.assert_array_equal(c.cpu().numpy()[:, 1:], c0)
    np.testing.assert_array_equal(n.cpu().numpy(), n0)
    shape = [61, 3008, 3008]
    assert [g + (1 if i == 0 else 0) for i, g in enumerate(vz.grid_zyx)] == shape
    index = ops.SiteIndex.build(c, 1, shape)
    nbr = ops.rulebook_subm(c, index)
    idx0 = np.concatenate([np.zeros((c0.shape[0], 1), np.int32), c0], 1)
    np.testing.assert_array_equal(nbr.cpu().numpy(), oracle.subm_rulebook(idx0, 1, shape, [3, 3, 3]))
    # gather-scatter leg: one 16->16 SubM conv over the level (HBM/latency-bound layer)
    rng = np.random.default_rng(1)
    feat = rng.normal(size=(c0.shape[0], 16)).astype(np.float32)
    w = (rng.normal(size=(16, 3, 3, 3, 16)) * 0.1).astype(np.float32)
    w_kio = torch.from_numpy(w).reshape(16, -1, 16).permute(1, 2, 0).contiguous().cuda()
    got = ops.gather_conv(torch.from_numpy(feat).cuda(), 16, ops.pack_weight(w_kio), nbr, 27, c0.shape[0], 16).cpu().numpy()
    np.testing.assert_allclose(got, oracle.sparse_conv(feat, w, None, oracle.subm_rulebook(idx0, 1, shape, [3, 3, 3])), atol=1e-4)
