#!/usr/bin/env python3
"""time of ops.order_rows_by_taps (pattern pass + chunk sort) on the strided levels of FRAMES frames.  CPD_HIP_LIB=... FRAMES=48 python tools/order_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import ops
from cpd_amd.synthetic import WAYMO, waymo_cloud
B = int(os.environ.get("FRAMES", "48"))
DOWN = [([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [1, 1, 1]), ([3, 3, 3], [2, 2, 2], [0, 1, 1])]
vox = ops.Voxelizer(WAYMO["voxel_size"], WAYMO["point_cloud_range"], 5, 5, 1000000)
pts = [torch.from_numpy(waymo_cloud(s % 8)).cuda() for s in range(B)]
_, coords, _, _, nvox, index = vox.batch(pts, index_z_extra=1, canonical=True)
coords = coords[:int(nvox[B])]
shape = index.shape
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for k, s, p in DOWN:
    coords, index, shape = ops.conv_outset(coords, B, shape, k, s, p)
    print("level %s: %d rows  order_rows_by_taps(4096) %.0f us" % (shape, coords.shape[0], timeit(lambda: ops.order_rows_by_taps(coords, index, chunk_rows=4096))))
