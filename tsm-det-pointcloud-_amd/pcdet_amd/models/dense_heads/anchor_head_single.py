"""AnchorHeadSingle (reference pcdet/models/dense_heads/anchor_head_single.py:8-88): three 1x1 convs (class, box,
direction) + anchor decode.  Reads `encoded_bev_features` (list, the fork's key) or `spatial_features_2d`."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .anchor_head_template import AnchorHeadTemplate


_HYBRID = os.environ.get("SPX_HEAD_HYBRID", "1") != "0"      # dev knob


class _PointwiseHeadsFn(torch.autograd.Function):
    """1x1 convolution of a channels_last map as the three GEMM-shaped pieces that are fastest on MI355X for a 512 -> 72
    head (tools/dense_tail_probe.py, 4 x 200 x 176 pixels): forward = one GEMM over pixels (155 us; MIOpen's conv 267),
    input gradient = one GEMM (dy @ W), weight / bias gradient = MIOpen's conv backward-weights (which reduces over the
    141 k pixels better than a GEMM with that K).  fwd+bwd 489 us against 666 for conv2d's own backward.  Same dot
    products as the three 1x1 convs of reference anchor_head_single.py:54-68."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x.permute(0, 2, 3, 1), w.flatten(1), b)           # [N, H, W, Co]

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        n, h, wd, co = dy.shape
        dx = (dy.view(-1, co) @ w.flatten(1)).view(n, h, wd, x.shape[1]).permute(0, 3, 1, 2)
        _, dw, db = torch.ops.aten.convolution_backward(dy.permute(0, 3, 1, 2), x, w, [co], [1, 1], [0, 0], [1, 1], False,
                                                        [0, 0], 1, [False, True, True])
        return dx, dw, db


class AnchorHeadSingle(AnchorHeadTemplate):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range,
                 predict_boxes_when_training=True, **kwargs):
        super().__init__(model_cfg=model_cfg, num_class=num_class, class_names=class_names, grid_size=grid_size,
                         point_cloud_range=point_cloud_range, predict_boxes_when_training=predict_boxes_when_training)
        self.num_anchors_per_location = sum(self.num_anchors_per_location)
        self.conv_cls = nn.Conv2d(input_channels, self.num_anchors_per_location * self.num_class, kernel_size=1)
        self.conv_box = nn.Conv2d(input_channels, self.num_anchors_per_location * self.box_coder.code_size,
                                  kernel_size=1)
        if self.model_cfg.get('USE_DIRECTION_CLASSIFIER', None) is not None:
            self.conv_dir_cls = nn.Conv2d(input_channels, self.num_anchors_per_location * self.model_cfg.NUM_DIR_BINS,
                                          kernel_size=1)
        else:
            self.conv_dir_cls = None
        self.init_weights()

    def init_weights(self):
        pi = 0.01
        nn.init.constant_(self.conv_cls.bias, -np.log((1 - pi) / pi))
        nn.init.normal_(self.conv_box.weight, mean=0, std=0.001)

    def _heads(self, x):
        """The three 1x1 convs of reference anchor_head_single.py:54-68 as ONE conv over the concatenated filters: the
        512-channel map is read once (and its gradient produced by one dgrad instead of three plus two adds).  Every
        output channel is the same dot product as in the separate convs; parameters stay `conv_cls/conv_box/
        conv_dir_cls` (state_dict compatible)."""
        convs = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
        w = torch.cat([c.weight for c in convs], 0)
        b = torch.cat([c.bias for c in convs], 0)
        if (not x.is_contiguous(memory_format=torch.channels_last) or torch.is_autocast_enabled()
                or (torch.is_grad_enabled() and not _HYBRID)):
            y = F.conv2d(x, w, b).permute(0, 2, 3, 1)
        elif not torch.is_grad_enabled():
            y = F.linear(x.permute(0, 2, 3, 1), w.flatten(1), b)          # inference: a plain GEMM over pixels
        else:
            y = _PointwiseHeadsFn.apply(x, w, b)
        outs = [t.contiguous() for t in torch.split(y, [c.out_channels for c in convs], dim=3)]
        return outs[0], outs[1], (outs[2] if len(outs) > 2 else None)

    def forward(self, data_dict):
        if data_dict.get('encoded_bev_features', None) is not None:
            feats = data_dict['encoded_bev_features']
            x = feats[0] if len(feats) == 1 else torch.cat(feats, dim=1)
        else:
            x = data_dict['spatial_features_2d']
        x = x.float()
        cls_preds, box_preds, dir_cls_preds = self._heads(x)             # each [N, H, W, C]
        self.forward_ret_dict['cls_preds'] = cls_preds
        self.forward_ret_dict['box_preds'] = box_preds
        if dir_cls_preds is not None:
            self.forward_ret_dict['dir_cls_preds'] = dir_cls_preds
        if self.training:
            self.forward_ret_dict.update(self.assign_targets(gt_boxes=data_dict['gt_boxes']))
        if not self.training or self.predict_boxes_when_training:
            batch_cls_preds, batch_box_preds = self.generate_predicted_boxes(
                batch_size=data_dict['batch_size'], cls_preds=cls_preds, box_preds=box_preds,
                dir_cls_preds=dir_cls_preds)
            data_dict['batch_cls_preds'] = batch_cls_preds
            data_dict['batch_box_preds'] = batch_box_preds
            data_dict['cls_preds_normalized'] = False
        return data_dict
