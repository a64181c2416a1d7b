"""`spconv.utils` / `cumm.tensorview` names used by cpd/datasets/processor/data_processor.py:8-25,53."""
import types

import numpy as np
import torch

from .. import ops


class _TvTensor:
    """Just enough of cumm.tensorview.Tensor for VoxelGeneratorWrapper.generate (l.53-58)."""

    def __init__(self, arr):
        self._a = arr

    def numpy(self):
        return np.array(self._a, copy=True)

    def numpy_view(self):
        return self._a


tensorview = types.ModuleType("cumm.tensorview")
tensorview.Tensor = _TvTensor
tensorview.from_numpy = lambda a: _TvTensor(np.ascontiguousarray(a))


class Point2VoxelCPU3d:
    """[SPCONV] Point2VoxelCPU3d(vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels,
    max_num_points_per_voxel).point_to_voxel(tv_points) -> (voxels, coordinates(z,y,x), num_points).
    Same serial first-appearance semantics, computed by cpd_voxelize on the GPU."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel,
                 device="cuda"):
        self._vz = ops.Voxelizer(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
                                 max_num_voxels, device=device)
        self.grid_size = self._vz.grid_zyx[::-1]

    def point_to_voxel(self, pc):
        pts = pc.numpy_view() if isinstance(pc, _TvTensor) else np.asarray(pc)
        pts = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).to(self._vz.device)
        v, c, n, _, _ = self._vz(pts, coord_cols=3, want_voxels=True, want_mean=False)
        return _TvTensor(v.cpu().numpy()), _TvTensor(c.cpu().numpy()), _TvTensor(n.cpu().numpy())
