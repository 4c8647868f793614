"""Score threshold + rotated-BEV NMS selection (reference pcdet/models/model_utils/model_nms_utils.py:6-87)."""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    src_box_scores = box_scores
    if score_thresh is not None:
        scores_mask = box_scores >= score_thresh
        box_scores = box_scores[scores_mask]
        box_preds = box_preds[scores_mask]
    selected = []
    if box_scores.shape[0] > 0:
        top_scores, indices = torch.topk(box_scores, k=min(nms_config.NMS_PRE_MAXSIZE, box_scores.shape[0]))
        keep_idx, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(box_preds[indices][:, 0:7], top_scores,
                                                                    nms_config.NMS_THRESH)
        selected = indices[keep_idx[:nms_config.NMS_POST_MAXSIZE]]
    if score_thresh is not None:
        selected = scores_mask.nonzero().view(-1)[selected]
    return selected, src_box_scores[selected]


def multi_thresh(box_scores, box_labels, box_preds, nms_config, score_thresh=None):
    """Fork variant: per-class score threshold and NMS, then one more NMS across the survivors."""
    src_box_scores = box_scores
    selected, selected_end = [], []
    if score_thresh is not None:
        for i, cur_thresh in enumerate(score_thresh):
            cls_idx = ((i + 1) == box_labels).nonzero().view(-1)
            cur_scores = box_scores[cls_idx]
            keep = (cur_scores >= cur_thresh).nonzero().view(-1)
            cur_scores, cur_idx = cur_scores[keep], cls_idx[keep]
            if cur_scores.shape[0] > 0:
                top_scores, indices = torch.topk(cur_scores, k=min(nms_config.NMS_PRE_MAXSIZE, cur_scores.shape[0]))
                keep_idx, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(box_preds[cur_idx[indices]][:, 0:7],
                                                                            top_scores, nms_config.NMS_THRESH)
                selected.append(cur_idx[indices[keep_idx[:nms_config.NMS_POST_MAXSIZE]]])
    if len(selected):
        selected = torch.cat(selected, dim=0)
        keep_end, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(box_preds[selected][:, 0:7], box_scores[selected],
                                                                    nms_config.NMS_THRESH)
        selected_end = selected[keep_end]
    return selected_end, src_box_scores[selected_end]
