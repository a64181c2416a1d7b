"""Mirror of VoxelGeneratorWrapper (cpd/datasets/processor/data_processor.py:14-59): same constructor
keywords, `generate(points: np.ndarray) -> (voxels, coordinates, num_points)` numpy outputs, computed
by cpd_voxelize on the GPU (bit-identical to the serial CPU generator it replaces)."""
import numpy as np
import torch

from . import ops


class VoxelGeneratorWrapper:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels,
                 device="cuda"):
        self.spconv_ver = 2
        self._voxel_generator = ops.Voxelizer(vsize_xyz, coors_range_xyz, num_point_features,
                                              max_num_points_per_voxel, max_num_voxels, device=device)

    def generate(self, points):
        pts = torch.from_numpy(np.ascontiguousarray(points, np.float32)).to(self._voxel_generator.device)
        voxels, coordinates, num_points, _, _ = self._voxel_generator(pts, coord_cols=3, want_voxels=True, want_mean=False)
        return voxels.cpu().numpy(), coordinates.cpu().numpy(), num_points.cpu().numpy()

    def generate_device(self, points, batch_idx=0):
        """Device-resident variant for a GPU data path: torch tensor in, (voxels, voxel_coords[b,z,y,x],
        voxel_num_points, voxel_features) out -- what collate_batch (dataset.py:264), load_data_to_gpu
        (cpd/models/__init__.py:16-24) and MeanVFE (mean_vfe.py:41-43) produce together."""
        v, c, n, mean, _ = self._voxel_generator(points, batch_idx=batch_idx, coord_cols=4, want_voxels=True, want_mean=True)
        return v, c, n, mean
