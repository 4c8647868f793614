"""MeanVFE on libspx (reference pcdet/models/backbones_3d/vfe/mean_vfe.py:6-31).

Two entry modes, same outputs (`voxel_features [N,C]`, and `voxel_coords [N,4]` when it voxelises itself):
  * batch_dict already holds `voxels` / `voxel_num_points` (the reference's CPU-dataloader contract):
    one spx_mean_vfe kernel;
  * batch_dict holds only `points [N, 1+C]` (frame index in column 0, as collate_batch builds it,
    pcdet/datasets/dataset.py:173-178): the GPU voxeliser runs here with MeanVFE fused into its gather —
    the PointToVoxel scatter of the north-star path.  Needs VOXELIZE in model_cfg or dataset info in kwargs.
"""
import torch

from spx import ops

from .vfe_template import VFETemplate


class MeanVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size=None, point_cloud_range=None, grid_size=None,
                 **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features
        self.voxel_size = None if voxel_size is None else [float(v) for v in voxel_size]
        self.point_cloud_range = None if point_cloud_range is None else [float(v) for v in point_cloud_range]
        vcfg = model_cfg.get("VOXELIZE", None) if hasattr(model_cfg, "get") else None
        self.max_points = int(vcfg.get("MAX_POINTS_PER_VOXEL", 5)) if vcfg else 5
        mv = vcfg.get("MAX_NUMBER_OF_VOXELS", {"train": 16000, "test": 40000}) if vcfg else {"train": 16000,
                                                                                                "test": 40000}
        self.max_voxels = dict(mv) if isinstance(mv, dict) else {"train": int(mv), "test": int(mv)}
        self.keep_voxels = bool(vcfg.get("KEEP_VOXELS", False)) if vcfg else False

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        if "voxels" in batch_dict:
            voxels, num = batch_dict["voxels"], batch_dict["voxel_num_points"]
            batch_dict["voxel_features"] = ops.mean_vfe(voxels, num)
            return batch_dict
        points = batch_dict["points"]
        mode = "train" if self.training else "test"
        static = batch_dict.get("static_caps", None) is not None      # hipGraph mode: no host sync, rows at capacity
        out = ops.voxelize(points, self.point_cloud_range, self.voxel_size, self.max_points, self.max_voxels[mode],
                           batch_size=int(batch_dict["batch_size"]), batch_col=0, xyz_col=1, feat_col=1,
                           num_features=self.num_point_features, want_voxels=self.keep_voxels and not static,
                           sync=not static)
        if static:
            batch_dict["voxel_num_valid"] = out["d_num_voxels"]
        batch_dict["voxel_features"] = out["mean"]
        batch_dict["voxel_coords"] = out["coords"]
        batch_dict["voxel_num_points"] = out["num_points"]
        if self.keep_voxels:
            batch_dict["voxels"] = out["voxels"]
        return batch_dict
