"""AnchorHeadTemplate (API and arithmetic of reference pcdet/models/dense_heads/anchor_head_template.py:11-276):
anchors, target assignment, focal / smooth-L1 / direction losses, box decoding.  Device agnostic (anchors are
buffers), and the loss dictionary holds 0-d tensors instead of .item() floats so no host sync happens per step."""
import numpy as np
import torch
import torch.nn as nn

from ...utils import box_coder_utils, common_utils, loss_utils
from .target_assigner.anchor_generator import AnchorGenerator
from .target_assigner.axis_aligned_target_assigner import AxisAlignedTargetAssigner


class _FusedAnchorLoss(torch.autograd.Function):
    """losses[3] = (cls, loc, dir), each already weighted and divided by the batch size; one libspx launch group computes
    them together with d(loss)/d(prediction) (csrc/anchor_loss.hip), backward only scales the saved gradients."""

    @staticmethod
    def forward(ctx, cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, consts):
        from spx import ops
        dir_offset, cls_w, loc_w, dir_w, beta, alpha = consts
        losses, dcls, dbox, ddir = ops.anchor_loss(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors,
                                                   dir_offset, cls_w, loc_w, dir_w, beta, alpha)
        ctx.has_dir = ddir is not None
        ctx.save_for_backward(dcls, dbox, ddir if ddir is not None else dcls.new_empty(0))
        ctx.shapes = (cls_preds.shape, box_preds.shape, None if dir_preds is None else dir_preds.shape)
        return losses

    @staticmethod
    def backward(ctx, g):
        dcls, dbox, ddir = ctx.saved_tensors
        sc, sb, sd = ctx.shapes
        return ((dcls * g[0]).view(sc), (dbox * g[1]).view(sb), (ddir * g[2]).view(sd) if ctx.has_dir else None,
                None, None, None, None)


class AnchorHeadTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, class_names, grid_size, point_cloud_range, predict_boxes_when_training):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.class_names = class_names
        self.predict_boxes_when_training = predict_boxes_when_training
        self.use_multihead = self.model_cfg.get('USE_MULTIHEAD', False)

        anchor_target_cfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        self.box_coder = getattr(box_coder_utils, anchor_target_cfg.BOX_CODER)(
            num_dir_bins=anchor_target_cfg.get('NUM_DIR_BINS', 6), **anchor_target_cfg.get('BOX_CODER_CONFIG', {}))

        anchors, self.num_anchors_per_location = self.generate_anchors(
            self.model_cfg.ANCHOR_GENERATOR_CONFIG, grid_size=grid_size, point_cloud_range=point_cloud_range,
            anchor_ndim=self.box_coder.code_size)
        self._n_anchor_sets = len(anchors)
        for i, a in enumerate(anchors):
            self.register_buffer('_anchors_%d' % i, a, persistent=False)
        self.target_assigner = self.get_target_assigner(anchor_target_cfg)
        self.forward_ret_dict = {}
        self.build_losses(self.model_cfg.LOSS_CONFIG)

    @property
    def anchors(self):
        return [getattr(self, '_anchors_%d' % i) for i in range(self._n_anchor_sets)]

    @staticmethod
    def generate_anchors(anchor_generator_cfg, grid_size, point_cloud_range, anchor_ndim=7):
        gen = AnchorGenerator(anchor_range=point_cloud_range, anchor_generator_config=anchor_generator_cfg)
        gs = np.asarray(grid_size)
        feature_map_size = [gs[:2] // c['feature_map_stride'] for c in anchor_generator_cfg]
        anchors_list, per_location = gen.generate_anchors(feature_map_size)
        if anchor_ndim != 7:
            anchors_list = [torch.cat((a, a.new_zeros([*a.shape[0:-1], anchor_ndim - 7])), dim=-1)
                            for a in anchors_list]
        return anchors_list, per_location

    def get_target_assigner(self, anchor_target_cfg):
        if anchor_target_cfg.NAME == 'AxisAlignedTargetAssigner':
            return AxisAlignedTargetAssigner(model_cfg=self.model_cfg, class_names=self.class_names,
                                             box_coder=self.box_coder, match_height=anchor_target_cfg.MATCH_HEIGHT)
        raise NotImplementedError(anchor_target_cfg.NAME)

    def build_losses(self, losses_cfg):
        self.add_module('cls_loss_func', loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0))
        reg_loss_name = losses_cfg.get('REG_LOSS_TYPE', None) or 'WeightedSmoothL1Loss'
        self.add_module('reg_loss_func',
                        getattr(loss_utils, reg_loss_name)(code_weights=losses_cfg.LOSS_WEIGHTS['code_weights']))
        self.add_module('dir_loss_func', loss_utils.WeightedCrossEntropyLoss())

    def assign_targets(self, gt_boxes):
        """gt_boxes (B, M, 8)"""
        return self.target_assigner.assign_targets(self.anchors, gt_boxes)

    def _flat_anchors(self):
        return torch.cat(self.anchors, dim=-3)   # [1, H, W, sum(size), rot, 7]

    def get_cls_layer_loss(self):
        cls_preds = self.forward_ret_dict['cls_preds']
        labels = self.forward_ret_dict['box_cls_labels']
        batch_size = int(cls_preds.shape[0])
        cared = labels >= 0
        positives = labels > 0
        negatives = labels == 0
        cls_weights = (negatives * 1.0 + 1.0 * positives).float()
        if self.num_class == 1:
            labels = torch.where(positives, torch.ones_like(labels), labels)
        pos_normalizer = positives.sum(1, keepdim=True).float()
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        cls_targets = (labels * cared.type_as(labels)).long()
        one_hot = torch.zeros(*cls_targets.shape, self.num_class + 1, dtype=cls_preds.dtype, device=cls_targets.device)
        one_hot.scatter_(-1, cls_targets.unsqueeze(-1), 1.0)
        cls_preds = cls_preds.view(batch_size, -1, self.num_class)
        loss_src = self.cls_loss_func(cls_preds, one_hot[..., 1:], weights=cls_weights)
        cls_loss = loss_src.sum() / batch_size * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['cls_weight']
        return cls_loss, {'rpn_loss_cls': cls_loss.detach()}

    @staticmethod
    def add_sin_difference(boxes1, boxes2, dim=6):
        assert dim != -1
        s1, c1 = torch.sin(boxes1[..., dim:dim + 1]), torch.cos(boxes1[..., dim:dim + 1])
        s2, c2 = torch.sin(boxes2[..., dim:dim + 1]), torch.cos(boxes2[..., dim:dim + 1])
        b1 = torch.cat([boxes1[..., :dim], s1 * c2, boxes1[..., dim + 1:]], dim=-1)
        b2 = torch.cat([boxes2[..., :dim], c1 * s2, boxes2[..., dim + 1:]], dim=-1)
        return b1, b2

    @staticmethod
    def get_direction_target(anchors, reg_targets, one_hot=True, dir_offset=0, num_bins=2):
        batch_size = reg_targets.shape[0]
        anchors = anchors.view(batch_size, -1, anchors.shape[-1])
        rot_gt = reg_targets[..., 6] + anchors[..., 6]
        offset_rot = common_utils.limit_period(rot_gt - dir_offset, 0, 2 * np.pi)
        dir_cls = torch.floor(offset_rot / (2 * np.pi / num_bins)).long().clamp(min=0, max=num_bins - 1)
        if one_hot:
            t = torch.zeros(*dir_cls.shape, num_bins, dtype=anchors.dtype, device=dir_cls.device)
            t.scatter_(-1, dir_cls.unsqueeze(-1), 1.0)
            return t
        return dir_cls

    def get_box_reg_layer_loss(self):
        box_preds = self.forward_ret_dict['box_preds']
        dir_preds = self.forward_ret_dict.get('dir_cls_preds', None)
        reg_targets = self.forward_ret_dict['box_reg_targets']
        labels = self.forward_ret_dict['box_cls_labels']
        batch_size = int(box_preds.shape[0])
        positives = labels > 0
        reg_weights = positives.float()
        reg_weights = reg_weights / torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
        anchors = self._flat_anchors()
        anchors = anchors.view(1, -1, anchors.shape[-1]).repeat(batch_size, 1, 1)
        box_preds = box_preds.view(batch_size, -1, box_preds.shape[-1] // self.num_anchors_per_location)
        preds_sin, targets_sin = self.add_sin_difference(box_preds, reg_targets)
        loc_loss = self.reg_loss_func(preds_sin, targets_sin, weights=reg_weights).sum() / batch_size
        loc_loss = loc_loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['loc_weight']
        box_loss = loc_loss
        tb_dict = {'rpn_loss_loc': loc_loss.detach()}
        if dir_preds is not None:
            dir_targets = self.get_direction_target(anchors, reg_targets, dir_offset=self.model_cfg.DIR_OFFSET,
                                                    num_bins=self.model_cfg.NUM_DIR_BINS)
            dir_logits = dir_preds.view(batch_size, -1, self.model_cfg.NUM_DIR_BINS)
            weights = positives.type_as(dir_logits)
            weights = weights / torch.clamp(weights.sum(-1, keepdim=True), min=1.0)
            dir_loss = self.dir_loss_func(dir_logits, dir_targets, weights=weights).sum() / batch_size
            dir_loss = dir_loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['dir_weight']
            box_loss = box_loss + dir_loss
            tb_dict['rpn_loss_dir'] = dir_loss.detach()
        return box_loss, tb_dict

    def _unit_code_weights(self, rl):
        """code_weights are a constant of the config: read them once, not per step (a host read of device memory would
        put a sync in every training step and cannot be captured in a hipGraph)."""
        cw = rl.code_weights
        if cw is None:
            return True
        key = (cw.data_ptr(), cw._version)
        if getattr(self, '_cw_key', None) != key:
            self._cw_key, self._cw_unit = key, bool((cw == 1).all())
        return self._cw_unit

    def _fused_loss_ok(self):
        fr = self.forward_ret_dict
        rl = self.reg_loss_func
        return (fr['cls_preds'].is_cuda and type(rl) is loss_utils.WeightedSmoothL1Loss and self.num_class <= 8
                and fr['box_cls_labels'].dtype == torch.int32 and self.box_coder.code_size == 7
                and self._unit_code_weights(rl)
                and not self.use_multihead and self.num_class > 1)

    def get_loss_fused(self):
        """Same three losses as get_cls_layer_loss + get_box_reg_layer_loss, computed (with their gradients) by libspx."""
        fr = self.forward_ret_dict
        cls_preds, box_preds = fr['cls_preds'], fr['box_preds']
        dir_preds = fr.get('dir_cls_preds', None)
        b = int(cls_preds.shape[0])
        anchors = self._flat_anchors().reshape(-1, 7)
        a = anchors.shape[0]
        w = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        consts = (float(self.model_cfg.get('DIR_OFFSET', 0.0)), float(w['cls_weight']), float(w['loc_weight']),
                  float(w.get('dir_weight', 0.0)), float(self.reg_loss_func.beta), float(self.cls_loss_func.alpha))
        assert self.cls_loss_func.gamma == 2.0
        losses = _FusedAnchorLoss.apply(cls_preds.view(b, a, self.num_class), box_preds.view(b, a, 7),
                                        None if dir_preds is None else dir_preds.view(b, a, -1), fr['box_cls_labels'],
                                        fr['box_reg_targets'], anchors, consts)
        tb_dict = {'rpn_loss_cls': losses[0].detach(), 'rpn_loss_loc': losses[1].detach()}
        if dir_preds is not None:
            tb_dict['rpn_loss_dir'] = losses[2].detach()
            rpn_loss = losses.sum()
        else:
            rpn_loss = losses[0] + losses[1]
        tb_dict['rpn_loss'] = rpn_loss.detach()
        return rpn_loss, tb_dict

    def get_loss(self):
        if self._fused_loss_ok():
            return self.get_loss_fused()
        return self.get_loss_torch()

    def get_loss_torch(self):
        cls_loss, tb_dict = self.get_cls_layer_loss()
        box_loss, tb_box = self.get_box_reg_layer_loss()
        tb_dict.update(tb_box)
        rpn_loss = cls_loss + box_loss
        tb_dict['rpn_loss'] = rpn_loss.detach()
        return rpn_loss, tb_dict

    def generate_predicted_boxes(self, batch_size, cls_preds, box_preds, dir_cls_preds=None):
        """cls_preds (N,H,W,C1), box_preds (N,H,W,C2), dir (N,H,W,C3) -> (B, A, n_cls) logits, (B, A, 7) boxes."""
        anchors = self._flat_anchors()
        num_anchors = anchors.view(-1, anchors.shape[-1]).shape[0]
        batch_anchors = anchors.view(1, -1, anchors.shape[-1]).repeat(batch_size, 1, 1)
        batch_cls_preds = cls_preds.view(batch_size, num_anchors, -1).float()
        batch_box_preds = self.box_coder.decode_torch(box_preds.view(batch_size, num_anchors, -1), batch_anchors)
        if dir_cls_preds is not None:
            dir_offset = self.model_cfg.DIR_OFFSET
            dir_limit_offset = self.model_cfg.DIR_LIMIT_OFFSET
            dir_labels = torch.max(dir_cls_preds.view(batch_size, num_anchors, -1), dim=-1)[1]
            period = 2 * np.pi / self.model_cfg.NUM_DIR_BINS
            dir_rot = common_utils.limit_period(batch_box_preds[..., 6] - dir_offset, dir_limit_offset, period)
            heading = dir_rot + dir_offset + period * dir_labels.to(batch_box_preds.dtype)
            batch_box_preds = torch.cat([batch_box_preds[..., :6], heading.unsqueeze(-1), batch_box_preds[..., 7:]], -1)
        return batch_cls_preds, batch_box_preds

    def forward(self, **kwargs):
        raise NotImplementedError
