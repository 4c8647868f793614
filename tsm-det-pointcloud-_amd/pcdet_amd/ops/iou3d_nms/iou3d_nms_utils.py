"""Rotated-BEV NMS entry points (reference pcdet/ops/iou3d_nms/iou3d_nms_utils.py:84-118).

SURVEY.md §8(f-1) marks NMS as the NEXT row after the sparse hot path; until its HIP kernel lands these raise,
loudly, rather than fall back to a CPU implementation."""


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    raise NotImplementedError("rotated BEV NMS (SURVEY.md §8 row f-1) is not built yet: the HIP kernel is the next "
                              "row after the sparse-conv hot path; there is deliberately no CPU fallback")


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    raise NotImplementedError("axis-aligned NMS (SURVEY.md §8 row f-1) is not built yet")
