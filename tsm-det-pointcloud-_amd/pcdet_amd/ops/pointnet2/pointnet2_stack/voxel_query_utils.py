"""voxel_query / VoxelQueryAndGrouping on libspx (reference
pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py:10-116).  Same call signatures and return values; the search is
csrc/voxel_query.hip (spx_voxel_query), the grouping is a torch gather."""
import torch
import torch.nn as nn

from spx import ops


def voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    """-> idx (M, nsample) int32 rows of `xyz` in each query's ball (empty balls: all 0), empty_ball_mask (M) bool,
    density (M, 1) = occupied cells scanned / cells scanned (reference VoxelQuery.forward :14-46)."""
    z_range, y_range, x_range = max_range
    idx, cnt = ops.voxel_query(new_xyz, xyz, new_coords, point_indices, nsample, radius, (z_range, y_range, x_range))
    empty_ball_mask = idx[:, 0] == -1
    idx[empty_ball_mask] = 0
    volume = (x_range * 2 + 1) * (y_range * 2 + 1) * (z_range * 2 + 1)
    density = cnt.view(-1, 1) / volume
    return idx, empty_ball_mask, density


def grouping_operation(features, features_batch_cnt, idx, idx_batch_cnt):
    """features (N1+N2.., C), idx (M1+M2.., nsample) frame-local rows -> (M1+M2.., C, nsample) (reference
    pointnet2_utils.py:48-82); differentiable through the gather."""
    offsets = torch.cumsum(features_batch_cnt, 0) - features_batch_cnt          # first row of every frame
    frame_of_query = torch.repeat_interleave(torch.arange(idx_batch_cnt.shape[0], device=idx.device), idx_batch_cnt.long())
    rows = idx.long() + offsets.long()[frame_of_query][:, None]
    return features[rows].permute(0, 2, 1).contiguous()


class VoxelQueryAndGrouping(nn.Module):
    def __init__(self, max_range, radius, nsample):
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, voxel2point_indices):
        assert xyz.shape[0] == xyz_batch_cnt.sum() and new_coords.shape[0] == new_xyz_batch_cnt.sum()
        idx, empty_ball_mask, density = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords,
                                                    voxel2point_indices)
        # the table holds GLOBAL rows; grouping_operation wants frame-local ones (reference :93-99)
        offsets = torch.cumsum(xyz_batch_cnt, 0) - xyz_batch_cnt
        frame_of_query = torch.repeat_interleave(torch.arange(new_xyz_batch_cnt.shape[0], device=idx.device),
                                                 new_xyz_batch_cnt.long())
        idx = idx - offsets[frame_of_query][:, None].to(idx.dtype)
        idx[empty_ball_mask] = 0
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty_ball_mask, density


def voxel_query_dilated(max_range, stride, former_radius, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    """-> idx, empty_ball_mask, density_score (M, 1) = min(occupied cells scanned / nsample, 1) (reference
    VoxelQueryDilated.forward, voxel_query_utils.py:119-158; kernel voxel_query_gpu.cu:125-215)."""
    idx, cnt, _filled = ops.voxel_query_dilated(new_xyz, xyz, new_coords, point_indices, nsample, former_radius, radius,
                                                tuple(max_range), tuple(stride))
    empty_ball_mask = idx[:, 0] == -1
    idx[empty_ball_mask] = 0
    density_score = torch.clamp(cnt.view(-1, 1) / nsample, max=1.0)
    return idx, empty_ball_mask, density_score


class VoxelQueryAndGroupingDilated(nn.Module):
    """reference voxel_query_utils.py:168-236.  The reference turns global rows into frame-local ones with
    `idx.view(batch_size, -1, nsample)`, i.e. it needs the same number of queries in every frame; the per-frame offsets
    used here give the same result in that case and stay correct for ragged frames."""

    def __init__(self, max_range, stride, former_radius, radius, nsample):
        super().__init__()
        self.max_range, self.stride, self.former_radius, self.radius, self.nsample = \
            max_range, stride, former_radius, radius, nsample

    def forward(self, new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, voxel2point_indices):
        assert xyz.shape[0] == xyz_batch_cnt.sum() and new_coords.shape[0] == new_xyz_batch_cnt.sum()
        idx, empty_ball_mask, density_score = voxel_query_dilated(
            self.max_range, self.stride, self.former_radius, self.radius, self.nsample, xyz, new_xyz, new_coords,
            voxel2point_indices)
        offsets = torch.cumsum(xyz_batch_cnt, 0) - xyz_batch_cnt
        frame_of_query = torch.repeat_interleave(torch.arange(new_xyz_batch_cnt.shape[0], device=idx.device),
                                                 new_xyz_batch_cnt.long())
        idx = idx - offsets[frame_of_query][:, None].to(idx.dtype)
        idx[empty_ball_mask] = 0
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty_ball_mask, density_score
