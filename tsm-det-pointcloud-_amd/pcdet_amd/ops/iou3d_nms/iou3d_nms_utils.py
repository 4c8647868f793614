"""Rotated-BEV IoU / NMS entry points with the reference's signatures
(pcdet/ops/iou3d_nms/iou3d_nms_utils.py:12-118), backed by libspx (csrc/nms.hip).  No CPU fallback."""
import torch

from spx import ops


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) BEV IoU   (reference :30-45 boxes_iou_bev)"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return ops.boxes_iou_bev(boxes_a, boxes_b)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) 3-D IoU = BEV overlap x height overlap / union volume   (reference :48-81)"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = ops.boxes_iou_bev(boxes_a, boxes_b, overlap_only=True)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _nms(boxes, scores, thresh, pre_maxsize, axis_aligned):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep, cnt = ops.nms_bev(boxes[order], thresh, axis_aligned=axis_aligned)
    return order[keep[:int(cnt.item())]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """(N,7) boxes, (N) scores -> indices of kept boxes in descending-score order   (reference :84-99)"""
    return _nms(boxes, scores, thresh, pre_maxsize, False)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """axis-aligned (heading ignored) variant   (reference :102-118)"""
    return _nms(boxes, scores, thresh, None, True)
