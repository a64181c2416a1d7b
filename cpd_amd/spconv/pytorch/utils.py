"""`from spconv.pytorch.utils import PointToVoxel, gather_features_by_pc_voxel_id`
(imported, unused, at cpd/models/backbones_3d/spconv_backbone.py:10)."""
import torch

from ... import ops


class PointToVoxel:
    """[SPCONV] PointToVoxel(vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels,
    max_num_points_per_voxel, device): device tensor in, (voxels, coords zyx, num_points) out."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel,
                 device=torch.device("cuda")):
        self._vz = ops.Voxelizer(vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
                                 max_num_voxels, device=device)

    def __call__(self, pc, clear_voxels=True, empty_mean=False):
        v, c, n, _, _ = self._vz(pc.float().contiguous(), coord_cols=3, want_voxels=True, want_mean=False)
        return v, c, n


def gather_features_by_pc_voxel_id(seg_res_features, pc_voxel_id, invalid_value=0):
    res = seg_res_features.new_full((pc_voxel_id.shape[0],) + tuple(seg_res_features.shape[1:]), invalid_value)
    valid = pc_voxel_id >= 0
    res[valid] = seg_res_features[pc_voxel_id[valid].long()]
    return res
