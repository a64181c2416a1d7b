#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats -- python tools/order_kernel_probe.py : order_rows_kernel<E> vs chunk size on a level-1-size site list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cpd_amd import ops
from cpd_amd.engine import ModelConfig
from cpd_amd.synthetic import waymo_cloud
B = 16
cfg = ModelConfig()
vox = ops.Voxelizer(cfg.voxel_size, cfg.point_cloud_range, 5, 5, cfg.max_voxels)
clouds = [torch.from_numpy(waymo_cloud(s % 8)).cuda() for s in range(B)]
_, coords, _, feats, nvox = vox.batch(clouds)
coords = coords[:int(nvox[B])]
o_idx, o_index, o_shape = ops.conv_outset(coords, B, cfg.sparse_shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
for chunk in (1024, 4096, 8192, 16384):
    for _ in range(5):
        ops.order_rows_by_taps(o_idx, o_index, chunk_rows=chunk)
torch.cuda.synchronize()
print("rows", o_idx.shape[0])
