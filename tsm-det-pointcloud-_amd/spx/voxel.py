"""Voxel generators with the spconv call signatures the reference uses.

PointToVoxel      GPU analogue (signature seen commented at pcdet/models/roi_heads/EPoint_RoI_Head.py:93-100).
Point2VoxelCPU3d  the class VoxelGeneratorWrapper instantiates (pcdet/datasets/processor/data_processor.py:26,
                  37-43) and calls as .point_to_voxel(tv.from_numpy(points)) (:55).  Here it runs the SAME HIP
                  voxeliser (H2D, kernels, D2H) — there is no CPU implementation in the product; it must therefore
                  be used from the main process, not from forked dataloader workers.
"""
import numpy as np
import torch

from . import ops


class PointToVoxel(object):
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel,
                 device=None):
        self.vsize = [float(v) for v in vsize_xyz]
        self.range = [float(v) for v in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_voxels = int(max_num_voxels)
        self.max_points = int(max_num_points_per_voxel)
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self.grid_size = [int(round((self.range[3 + j] - self.range[j]) / self.vsize[j])) for j in range(3)]

    def __call__(self, points, clear_voxels=True, empty_mean=False):
        """points [N, C] on the GPU -> (voxels [M,T,C], coords [M,3] (z,y,x), num_points [M])."""
        out = ops.voxelize(points.to(self.device), self.range, self.vsize, self.max_points, self.max_voxels,
                           batch_size=1, num_features=self.num_point_features, want_mean=False)
        return out["voxels"], out["coords"][:, 1:].contiguous(), out["num_points"]

    def generate_batch(self, points_with_batch_col, batch_size, want_voxels=True):
        """collate_batch-style points [N, 1+C] (leading frame index) -> dict incl. coords [M,4] and MeanVFE."""
        return ops.voxelize(points_with_batch_col.to(self.device), self.range, self.vsize, self.max_points,
                            self.max_voxels, batch_size=batch_size, batch_col=0, xyz_col=1, feat_col=1,
                            num_features=self.num_point_features, want_voxels=want_voxels)


class _TV(object):
    """Minimal stand-in for cumm.tensorview.Tensor: what data_processor.py:55-60 touches (.numpy())."""

    def __init__(self, arr):
        self._a = arr

    def numpy(self):
        return np.array(self._a)

    def numpy_view(self):
        return self._a


class Point2VoxelCPU3d(object):
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels=None,
                 max_num_points_per_voxel=None, **kw):
        self._gen = PointToVoxel(vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels,
                                 max_num_points_per_voxel)

    def point_to_voxel(self, pc, clear_voxels=True):
        arr = pc.numpy_view() if hasattr(pc, "numpy_view") else np.asarray(pc)
        v, c, n = self._gen(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)))
        return _TV(v.cpu().numpy()), _TV(c.cpu().numpy()), _TV(n.cpu().numpy())
