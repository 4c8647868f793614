"""DataProcessor — the per-sample point pipeline in front of the detector (reference
pcdet/datasets/processor/data_processor.py:15-226).

Same contract as the reference: built from the yaml list `DATA_PROCESSOR`, every entry names a method; calling a
method with `data_dict=None` binds its config and returns the callable that `forward` later applies in order.
`transform_points_to_voxels` voxelises through `VoxelGeneratorWrapper`, i.e. through `spconv.utils.Point2VoxelCPU3d`
— which in this package is libspx's GPU voxeliser behind the spconv 2.x host API (H2D, kernels, D2H): call it from the
main process.  `transform_points_to_voxels_placeholder` only computes the grid and leaves `points` for the on-device
path (MeanVFE / DynamicMeanVFE voxelise on the GPU inside the model), which is what the benchmark uses.
"""
from functools import partial

import numpy as np

import spx as spconv_pkg

from ...utils import box_utils, common_utils


class VoxelGeneratorWrapper(object):
    """spconv 2.x flavour of the reference wrapper (data_processor.py:15-61): generate(points) -> voxels [M, T, C],
    coordinates [M, 3] (z, y, x), num_points [M] as numpy arrays."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        self.spconv_ver = 2
        self._voxel_generator = spconv_pkg.Point2VoxelCPU3d(
            vsize_xyz=vsize_xyz, coors_range_xyz=coors_range_xyz, num_point_features=num_point_features,
            max_num_points_per_voxel=max_num_points_per_voxel, max_num_voxels=max_num_voxels)

    def generate(self, points):
        voxels, coordinates, num_points = self._voxel_generator.point_to_voxel(np.ascontiguousarray(points, np.float32))
        return voxels.numpy(), coordinates.numpy(), num_points.numpy()


class DataProcessor(object):
    def __init__(self, processor_configs, point_cloud_range, training, num_point_features):
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.training = training
        self.num_point_features = num_point_features
        self.mode = 'train' if training else 'test'
        self.grid_size = self.voxel_size = None
        self.voxel_generator = None
        self.data_processor_queue = [getattr(self, cfg.NAME)(config=cfg) for cfg in processor_configs]

    def _set_grid(self, voxel_size):
        extent = self.point_cloud_range[3:6] - self.point_cloud_range[0:3]
        self.grid_size = np.round(extent / np.asarray(voxel_size)).astype(np.int64)
        self.voxel_size = voxel_size

    def mask_points_and_boxes_outside_range(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.mask_points_and_boxes_outside_range, config=config)
        if data_dict.get('points', None) is not None:
            keep = common_utils.mask_points_by_range(data_dict['points'], self.point_cloud_range)
            data_dict['points'] = data_dict['points'][keep]
        if data_dict.get('gt_boxes', None) is not None and config.REMOVE_OUTSIDE_BOXES and self.training:
            keep = box_utils.mask_boxes_outside_range_numpy(data_dict['gt_boxes'], self.point_cloud_range,
                                                            min_num_corners=config.get('min_num_corners', 1))
            data_dict['gt_boxes'] = data_dict['gt_boxes'][keep]
        return data_dict

    def shuffle_points(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.shuffle_points, config=config)
        if config.SHUFFLE_ENABLED[self.mode]:
            pts = data_dict['points']
            data_dict['points'] = pts[np.random.permutation(pts.shape[0])]
        return data_dict

    def transform_points_to_voxels_placeholder(self, data_dict=None, config=None):
        if data_dict is None:                      # grid only; voxelisation happens on the device inside the model
            self._set_grid(config.VOXEL_SIZE)
            return partial(self.transform_points_to_voxels_placeholder, config=config)
        return data_dict

    def calculate_grid_size(self, data_dict=None, config=None):
        if data_dict is None:
            self._set_grid(config.VOXEL_SIZE)
            return partial(self.calculate_grid_size, config=config)
        return data_dict

    def transform_points_to_voxels(self, data_dict=None, config=None):
        if data_dict is None:
            self._set_grid(config.VOXEL_SIZE)
            return partial(self.transform_points_to_voxels, config=config)
        if self.voxel_generator is None:           # created lazily, as in the reference (:138-145)
            self.voxel_generator = VoxelGeneratorWrapper(
                vsize_xyz=config.VOXEL_SIZE, coors_range_xyz=self.point_cloud_range,
                num_point_features=self.num_point_features, max_num_points_per_voxel=config.MAX_POINTS_PER_VOXEL,
                max_num_voxels=config.MAX_NUMBER_OF_VOXELS[self.mode])
        voxels, coordinates, num_points = self.voxel_generator.generate(data_dict['points'])
        if not data_dict['use_lead_xyz']:
            voxels = voxels[..., 3:]               # drop xyz, keep the remaining point features (:151-152)
        data_dict['voxels'] = voxels
        data_dict['voxel_coords'] = coordinates
        data_dict['voxel_num_points'] = num_points
        return data_dict

    def forward(self, data_dict):
        for step in self.data_processor_queue:
            data_dict = step(data_dict=data_dict)
        return data_dict
