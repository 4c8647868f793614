"""Voxel-centroid aggregation (reference pcdet/utils/voxel_aggregation_utils.py:9-161).

`get_centroid_per_voxel` — group points by integer voxel index and average them — is the same operation as the
dynamic voxeliser (unique cells in lexicographic order + per-cell mean), so it runs on libspx's spx_dynamic_voxelize: the
index columns are fed as coordinates of a unit grid, the point columns ride along as features.  Deterministic summation
order (the reference's scatter_add_ on a GPU is not).  The other helpers are index arithmetic in stock torch ops.
"""
import torch

from spx import ops

from . import common_utils


def get_overlapping_voxel_indices(point_coords, downsample_times, voxel_size, point_cloud_range):
    """(N, 3) xyz -> (N, 3) voxel index (x, y, z) at `downsample_times` x the base voxel size, (-1, -1, -1) for points
    outside the grid (reference :9-45: true division then truncation by .long())."""
    assert point_coords.shape[1] == 3
    vs = torch.tensor(voxel_size, device=point_coords.device).float() * downsample_times
    rng = torch.tensor(point_cloud_range, device=point_coords.device).float()
    idx = (point_coords - rng[0:3]) / vs
    grid = ((rng[3:6] - rng[0:3]) / vs).long()
    outside = ((idx < 0) | (idx >= grid)).any(dim=-1)
    idx = torch.where(outside[:, None], torch.full_like(idx, -1.0), idx)
    return idx.long()


def get_centroid_per_voxel(points, voxel_idxs, num_points_in_voxel=None):
    """points (N, 4 + f) [bxyz + f], voxel_idxs (N, 4) non-negative ints -> centroids (N', 4 + f), their voxel indices
    (N', 4) in torch.unique(dim=0) order, rows merged per voxel (N', torch.unique's counts in both modes, as the
    reference returns them), and each point's voxel row (N)  (reference :132-161).
    With `num_points_in_voxel` the mean is weighted by it (centroids of centroids)."""
    assert points.shape[0] == voxel_idxs.shape[0]
    n, c = points.shape
    if n == 0:
        z = voxel_idxs.new_zeros((0,), dtype=torch.int64)
        return points.new_zeros((0, c)), voxel_idxs.new_zeros((0, 4)), z, z
    idx_f = voxel_idxs.to(torch.float32)
    extent = (voxel_idxs.amax(dim=0) + 1).tolist()           # one host sync: the unit grid that holds every index
    assert max(extent) < (1 << 24), "indices must be exactly representable in fp32"
    pts = points.to(torch.float32)
    if num_points_in_voxel is not None:
        w = num_points_in_voxel.to(torch.float32).unsqueeze(-1)
        payload = torch.cat((pts * w, w), dim=1)
    else:
        payload = pts
    table = torch.cat((idx_f, payload), dim=1).contiguous()   # [c0, c1, c2, c3, payload...]
    out = ops.dynamic_voxelize(table, [0.0, 0.0, 0.0, float(extent[1]), float(extent[2]), float(extent[3])],
                               [1.0, 1.0, 1.0], batch_size=int(extent[0]), batch_col=0, xyz_col=1)
    means = out["features"][:, 3:]                            # the three leading columns are the cell's own indices
    inverse = out["inverse"].long()
    counts = torch.bincount(inverse, minlength=means.shape[0])
    if num_points_in_voxel is not None:
        centroids = means[:, :c] / means[:, c:c + 1]          # (sum p*w / m) / (sum w / m)
    else:
        centroids = means
    cells = out["coords"][:, [0, 3, 2, 1]].to(voxel_idxs.dtype)   # kernel returns (c0, c3, c2, c1)
    return centroids, cells, counts, inverse


def get_nonempty_voxel_feature_indices(voxel_indices, x_conv):
    """Rows of the sparse tensor `x_conv` that sit at `voxel_indices` (N, 4) and the mask of indices that hit an active
    voxel (reference :103-129)."""
    table = common_utils.generate_voxel2pinds(x_conv)
    rows = table[voxel_indices[:, 0], voxel_indices[:, 1], voxel_indices[:, 2], voxel_indices[:, 3]]
    hit = rows != -1
    return rows[hit].long(), hit
