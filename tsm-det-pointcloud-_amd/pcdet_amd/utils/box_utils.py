"""Axis-aligned 'nearest BEV' IoU used by the anchor target assigner
(semantics of reference pcdet/utils/box_utils.py:249-298)."""
import numpy as np
import torch

from . import common_utils


def boxes_iou_normal(boxes_a, boxes_b):
    """(N,4),(M,4) [x1,y1,x2,y2] -> IoU (N,M); also broadcasts leading batch dims of boxes_b (B,M,4) -> (B,N,M)."""
    a = boxes_a[..., :, None, :]
    b = boxes_b[..., None, :, :]
    w = (torch.min(a[..., 2], b[..., 2]) - torch.max(a[..., 0], b[..., 0])).clamp_min(0)
    h = (torch.min(a[..., 3], b[..., 3]) - torch.max(a[..., 1], b[..., 1])).clamp_min(0)
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    inter = w * h
    return inter / (area_a + area_b - inter).clamp_min(1e-6)


def boxes3d_lidar_to_aligned_bev_boxes(boxes3d):
    """(...,7+C) lidar boxes -> (...,4) axis-aligned BEV boxes, swapping dx/dy when |heading| is nearer 90deg."""
    rot = common_utils.limit_period(boxes3d[..., 6], offset=0.5, period=np.pi).abs()
    dims = torch.where(rot[..., None] < np.pi / 4, boxes3d[..., [3, 4]], boxes3d[..., [4, 3]])
    return torch.cat((boxes3d[..., 0:2] - dims / 2, boxes3d[..., 0:2] + dims / 2), dim=-1)


def boxes3d_nearest_bev_iou(boxes_a, boxes_b):
    return boxes_iou_normal(boxes3d_lidar_to_aligned_bev_boxes(boxes_a), boxes3d_lidar_to_aligned_bev_boxes(boxes_b))


def boxes_to_corners_3d(boxes3d):
    """(N, 7) [x, y, z, dx, dy, dz, heading] -> (N, 8, 3) corners (reference pcdet/utils/box_utils.py:28-53, same corner
    numbering: 0-3 bottom face counter-clockwise from (+x, +y), 4-7 the top face above them).  numpy in, numpy out."""
    import numpy as np
    is_numpy = isinstance(boxes3d, np.ndarray)
    b = torch.as_tensor(boxes3d, dtype=torch.float32)
    signs = b.new_tensor([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                          [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]) * 0.5
    local = b[:, None, 3:6] * signs[None]                                   # (N, 8, 3) in the box frame
    cos, sin = torch.cos(b[:, 6])[:, None], torch.sin(b[:, 6])[:, None]
    x = local[..., 0] * cos - local[..., 1] * sin
    y = local[..., 0] * sin + local[..., 1] * cos
    corners = torch.stack([x, y, local[..., 2]], dim=-1) + b[:, None, 0:3]
    return corners.numpy() if is_numpy else corners


def mask_boxes_outside_range_numpy(boxes, limit_range, min_num_corners=1):
    """Keep boxes with at least `min_num_corners` corners inside the range (reference box_utils.py:56-72)."""
    import numpy as np
    limit_range = np.asarray(limit_range, dtype=np.float32)
    corners = boxes_to_corners_3d(np.asarray(boxes)[:, 0:7])
    inside = ((corners >= limit_range[0:3]) & (corners <= limit_range[3:6])).all(axis=2)
    return inside.sum(axis=1) >= min_num_corners
