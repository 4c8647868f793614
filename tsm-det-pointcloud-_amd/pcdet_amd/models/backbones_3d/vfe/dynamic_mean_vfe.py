"""DynamicMeanVFE — dynamic (cap-free) voxelisation + per-voxel mean on the GPU (SURVEY.md §8a row a5'; reference
pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:13-76).

Same constructor and batch_dict contract as the reference: reads `points` [N, 1 + C] = (batch_idx, x, y, z, ...), writes
`voxel_features` [M, C] (mean of (x, y, z, ...) over the voxel's points) and `voxel_coords` [M, 4] = (b, z, y, x) with the
voxels sorted by the reference's merge key b*XYZ + cx*YZ + cy*Z + cz.  The reference needs torch_scatter and sums with
float atomics; here libspx's spx_dynamic_voxelize does the whole thing (bitmap rank instead of torch.unique's sort, fixed
summation order)."""
import torch

from spx import ops

from .vfe_template import VFETemplate


class DynamicMeanVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.grid_size = [int(g) for g in grid_size]

    def get_output_feature_dim(self):
        return self.num_point_features

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        points = batch_dict['points']                      # (batch_idx, x, y, z, i, e)
        vox = ops.dynamic_voxelize(points, self.point_cloud_range, self.voxel_size, batch_size=batch_dict['batch_size'],
                                   batch_col=0, xyz_col=1)
        batch_dict['voxel_features'] = vox['features']
        batch_dict['voxel_coords'] = vox['coords']
        batch_dict['point_to_voxel'] = vox['inverse']     # torch.unique's inverse in the reference (unq_inv, :59)
        return batch_dict
