"""Synthetic point clouds and boxes (build-owned, seeded) for parity tests and bench.py.

W-cloud: a 64-beam x 2650-azimuth spinning lidar (elevations -17.6..+2.4 deg, sensor 2 m above a
ground plane at z=0) ray-cast against the ground and 60 vertical cylinders/"buildings", clipped at
75 m, 2 cm range noise, sub-sampled to the requested number of returns (160 k = the Waymo-shape
config of BASELINE.json). Features [x, y, z, intensity, elongation] f32.
K-cloud: the same scene model restricted to the KITTI field of view, [x, y, z, intensity].
"""
import numpy as np

# Config W (tools/cfgs/dataset_configs/waymo_unsupervised/waymo_unsupervised_cproto.yaml:118,166-172)
WAYMO = dict(point_cloud_range=[-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], voxel_size=[0.1, 0.1, 0.15],
             max_points_per_voxel=5, max_voxels=1000000, num_point_features=5)
# Config K (KITTI-shape, 0.05 m; height_compression.py:48 defaults)
KITTI = dict(point_cloud_range=[0.0, -40.0, -3.0, 70.4, 40.0, 1.0], voxel_size=[0.05, 0.05, 0.1],
             max_points_per_voxel=5, max_voxels=1000000, num_point_features=4)
# Config C1: KITTI range at 0.1 m voxels
KITTI_C1 = dict(point_cloud_range=[0.0, -40.0, -3.0, 70.4, 40.0, 1.0], voxel_size=[0.1, 0.1, 0.1],
                max_points_per_voxel=5, max_voxels=1000000, num_point_features=4)


def _raycast(rng, n_az, elev_deg, sensor_z, n_obj, max_range, fov=None):
    elev = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], 64))
    az = np.linspace(0.0, 2 * np.pi, n_az, endpoint=False)
    if fov is not None:
        az = np.linspace(fov[0], fov[1], n_az)
    e, a = np.meshgrid(elev, az, indexing="ij")
    d = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], -1).reshape(-1, 3)
    t_best = np.full(d.shape[0], np.inf)
    dz = d[:, 2]
    tg = np.where(dz < -1e-6, -sensor_z / np.minimum(dz, -1e-6), np.inf)
    t_best = np.minimum(t_best, tg)
    # objects: vertical cylinders (cars/pedestrians) and a ring of large "buildings"
    n_small = n_obj - 12
    cx = np.concatenate([rng.uniform(-70, 70, n_small), 62 * np.cos(np.linspace(0, 2 * np.pi, 12, endpoint=False))])
    cy = np.concatenate([rng.uniform(-70, 70, n_small), 62 * np.sin(np.linspace(0, 2 * np.pi, 12, endpoint=False))])
    rad = np.concatenate([rng.uniform(0.5, 2.5, n_small), rng.uniform(6.0, 12.0, 12)])
    hgt = np.concatenate([rng.uniform(1.2, 3.5, n_small), rng.uniform(6.0, 14.0, 12)])
    keep = np.hypot(cx, cy) > rad + 3.0   # nothing on top of the sensor
    dxy = d[:, :2]
    a2 = (dxy ** 2).sum(1)
    for k in np.nonzero(keep)[0]:
        c = np.array([cx[k], cy[k]])
        b = -(dxy @ c)
        cc = c @ c - rad[k] ** 2
        disc = b * b - a2 * cc
        ok = disc > 0
        t = np.where(ok, (-b - np.sqrt(np.where(ok, disc, 0))) / a2, np.inf)
        z = sensor_z + t * dz
        t = np.where((t > 0.5) & (z >= 0.0) & (z <= hgt[k]), t, np.inf)
        t_best = np.minimum(t_best, t)
    hit = np.isfinite(t_best) & (t_best < max_range)
    t = t_best[hit] + rng.normal(0, 0.02, hit.sum())
    pts = d[hit] * t[:, None]
    pts[:, 2] += sensor_z
    return pts


def waymo_cloud(seed=0, n_points=160000, n_az=2650):
    """Waymo-shape cloud, [n_points, 5] f32 in scan order."""
    rng = np.random.default_rng(seed)
    while True:
        pts = _raycast(rng, n_az, (-17.6, 2.4), 2.0, 60, 75.0)
        if pts.shape[0] >= n_points:
            break
        n_az = int(n_az * 1.15) + 1   # denser azimuth sampling until enough returns
    if pts.shape[0] > n_points:
        sel = np.sort(rng.choice(pts.shape[0], n_points, replace=False))
        pts = pts[sel]
    out = np.empty((pts.shape[0], 5), np.float32)
    out[:, :3] = pts
    out[:, 3] = rng.uniform(0, 1, pts.shape[0])
    out[:, 4] = rng.uniform(0, 1, pts.shape[0])
    return out


def kitti_cloud(seed=0, n_points=20000):
    """KITTI-shape cloud (front field of view), [n_points, 4] f32."""
    rng = np.random.default_rng(seed + 1000)
    n_az = 900
    while True:
        pts = _raycast(rng, n_az, (-24.8, 2.0), 1.73, 40, 80.0, fov=(-np.pi / 4, np.pi / 4))
        m = (pts[:, 0] > 0) & (pts[:, 0] < 70.4) & (np.abs(pts[:, 1]) < 40)
        pts = pts[m]
        pts[:, 2] -= 1.73  # KITTI velodyne frame: ground at z = -1.73
        if pts.shape[0] >= n_points:
            break
        n_az = int(n_az * 1.2) + 1
    if pts.shape[0] > n_points:
        sel = np.sort(rng.choice(pts.shape[0], n_points, replace=False))
        pts = pts[sel]
    out = np.empty((pts.shape[0], 4), np.float32)
    out[:, :3] = pts
    out[:, 3] = rng.uniform(0, 1, pts.shape[0])
    return out


def random_boxes(seed, n, span=75.0, dup_frac=0.1):
    """NMS test boxes [n,7] + distinct scores: centres U(-span, span), dims LogNormal around
    (4.7, 2.1, 1.7), heading U(-pi, pi), 10 % near-duplicates."""
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-span, span, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3:6] = np.exp(rng.normal(0, 0.25, (n, 3))) * np.array([4.7, 2.1, 1.7])
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    k = int(n * dup_frac)
    if k:
        b[n - k:] = b[:k]
        b[n - k:, :2] += rng.normal(0, 0.25, (k, 2)).astype(np.float32)
        b[n - k:, 6] += rng.normal(0, 0.05, k).astype(np.float32)
    scores = (rng.permutation(n).astype(np.float32) + 1) / (n + 1)
    return b.astype(np.float32), scores


# PredifinedSize of waymo_unsupervised_cproto.yaml:85 (Vehicle, Pedestrian, Cyclist)
PREDEFINED_SIZE = np.array([[5.065, 1.86, 1.49], [1.0, 1.0, 2.0], [1.9, 0.85, 1.8]], np.float32)


def gt_boxes(seed, n=30, span=70.0):
    """Synthetic ground truth of the config-3 train step: [n, 8] =
    (x, y, z, dx, dy, dz, heading, class 1..3), centres uniform within +-span on the ground,
    sizes PredifinedSize x U(0.9, 1.1), heading U(-pi, pi)."""
    rng = np.random.default_rng(1000 + seed)
    cls = rng.integers(1, 4, n)
    dims = PREDEFINED_SIZE[cls - 1] * rng.uniform(0.9, 1.1, (n, 3)).astype(np.float32)
    b = np.zeros((n, 8), np.float32)
    b[:, 0:2] = rng.uniform(-span, span, (n, 2))
    b[:, 2] = dims[:, 2] / 2.0                         # resting on the ground plane z = 0
    b[:, 3:6] = dims
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    b[:, 7] = cls
    return b
