"""[SPCONV] Point2VoxelCPU3d semantics checked against an independent
dict-based Python restatement on small clouds, incl. boundary points, caps and ragged inputs."""
import numpy as np
import pytest

VS = [0.1, 0.1, 0.15]
PCR = [-75.2, -75.2, -2, 75.2, 75.2, 4]


def py_voxelize(points, vs, pcr, P, maxv):
    vs = np.asarray(vs, np.float32); pcr = np.asarray(pcr, np.float32)
    grid = np.round((pcr[3:] - pcr[:3]) / vs).astype(np.int64)[::-1]
    table, coords, vox, num = {}, [], [], []
    for p in points:
        c = np.floor((p[:3][::-1] - pcr[:3][::-1]) / vs[::-1])      # fp32 sub/div, z,y,x
        if np.any(c < 0) or np.any(c >= grid) or not np.all(np.isfinite(c)):
            continue
        key = tuple(int(v) for v in c)
        v = table.get(key)
        if v is None:
            if len(coords) >= maxv:
                continue
            v = len(coords); table[key] = v
            coords.append(key); vox.append(np.zeros((P, points.shape[1]), np.float32)); num.append(0)
        if num[v] < P:
            vox[v][num[v]] = p; num[v] += 1
    if not coords:
        return (np.zeros((0, P, points.shape[1]), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32))
    return np.stack(vox), np.asarray(coords, np.int32), np.asarray(num, np.int32)


def cloud(rng, n, c=5, dense=False):
    pts = np.zeros((n, c), np.float32)
    span = 3.0 if dense else 80.0
    pts[:, 0] = rng.uniform(-span, span, n); pts[:, 1] = rng.uniform(-span, span, n)
    pts[:, 2] = rng.uniform(-2.5, 4.5, n); pts[:, 3:] = rng.uniform(0, 1, (n, c - 3))
    return pts


@pytest.mark.parametrize("n,dense,P,maxv", [(3000, False, 5, 100000), (4000, True, 5, 100000), (4000, True, 3, 500),
                                            (0, False, 5, 10), (1, False, 5, 10)])
def test_voxelize_matches_python(oracle, n, dense, P, maxv):
    rng = np.random.default_rng(n + P)
    pts = cloud(rng, n, dense=dense)
    if n > 10:   # boundary cases: exactly on lower / upper bounds, nan, huge
        pts[0, :3] = [-75.2, -75.2, -2.0]; pts[1, :3] = [75.2, 0, 0]; pts[2, :3] = [0, 0, 4.0]
        pts[3, 0] = np.nan; pts[4, 1] = 1e30; pts[5, :3] = [75.19999, 75.19999, 3.99999]
    v, c, k = oracle.voxelize(pts, VS, PCR, P, maxv)
    v2, c2, k2 = py_voxelize(pts, VS, PCR, P, maxv)
    np.testing.assert_array_equal(c, c2); np.testing.assert_array_equal(k, k2); np.testing.assert_array_equal(v, v2)
    assert oracle.grid_size(VS, PCR) == [40, 1504, 1504]
