"""CPU: the oracle's restatement of the RoI-pooling kernels against the fixture produced by running the
reference's own Python pooling stack on top of it (tests/golden/make_golden.py section 8)."""
import numpy as np


def test_voxel2pinds_and_query_reproduce_reference_python_outputs(oracle, golden):
    g = golden("roi_pool")
    cells, shape = g["cells"], [int(v) for v in g["shape"]]
    v2p = oracle.voxel2pinds(cells, 2, shape)
    assert v2p.shape == (2, *shape) and (v2p >= 0).sum() == cells.shape[0]
    b, z, y, x = cells[100]
    assert v2p[b, z, y, x] == 100
    nc = g["new_coords_bxyz"][:, [0, 3, 2, 1]]
    raw = oracle.voxel_query([1, 2, 2], 0.9, 8, g["xyz"], g["grid_xyz"].reshape(-1, 3), nc, v2p)
    empty = raw[:, 0] == -1
    np.testing.assert_array_equal(empty, g["query_empty"])
    raw[empty] = 0                                               # VoxelQuery.forward, voxel_query_utils.py:38-39
    np.testing.assert_array_equal(raw, g["query_idx"])
    # every returned neighbour really is within the radius and inside the scan window
    xyz, q = g["xyz"], g["grid_xyz"].reshape(-1, 3)
    d = np.linalg.norm(xyz[raw[~empty]] - q[~empty][:, None, :], axis=-1)
    assert (d <= 0.9 + 1e-6).all()


def test_group_points_matches_indexing(oracle):
    rng = np.random.default_rng(0)
    feat = rng.normal(size=(50, 6)).astype(np.float32)
    fcnt, icnt = np.array([20, 30], np.int32), np.array([7, 5], np.int32)
    idx = np.concatenate([rng.integers(0, 20, (7, 4)), rng.integers(0, 30, (5, 4))]).astype(np.int32)
    out = oracle.group_points(feat, fcnt, idx, icnt)
    want = np.concatenate([feat[idx[:7]], feat[20 + idx[7:]]]).transpose(0, 2, 1)
    np.testing.assert_array_equal(out, want)
