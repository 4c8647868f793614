"""AxisAlignedTargetAssigner — result-identical, BATCHED restatement of the reference's per-sample / per-class
Python loops (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-210).

The reference loops `for k in range(batch)`, `for anchor class`, builds index lists with nonzero() and syncs with
the host several times per (sample, class).  Here one class is assigned for ALL samples at once with masks:
no nonzero(), no .item(), no host sync — the rule per anchor a (gt set G of this class in this frame) is

    m = max_j IoU(a, j), g = argmax_j (first max);  forced(a) = exists j: IoU(a,j) == max_a' IoU(a',j) > 0
    label  = class(g)  if m >= matched_thr or forced(a)
           = 0         if m <  unmatched_thr (and not forced)        [also when G is empty]
           = -1        otherwise (ignored)
    target = encode(gt[g], a) for label > 0, else 0;   reg_weight = 1 for label > 0

which is what assign_targets_single computes with POS_FRACTION < 0 (sampling disabled), NORM_BY_NUM_EXAMPLES False.
Padding rule of the reference kept: trailing all-zero gt rows are dropped (the first row always stays), and a
class id of 0 indexes class_names[-1].
"""
import torch

from ....utils import box_utils


class AxisAlignedTargetAssigner(object):
    def __init__(self, model_cfg, class_names, box_coder, match_height=False):
        anchor_generator_cfg = model_cfg.ANCHOR_GENERATOR_CONFIG
        anchor_target_cfg = model_cfg.TARGET_ASSIGNER_CONFIG
        self.box_coder = box_coder
        self.match_height = match_height
        self.class_names = list(class_names)
        self.anchor_class_names = [c['class_name'] for c in anchor_generator_cfg]
        self.pos_fraction = anchor_target_cfg.POS_FRACTION if anchor_target_cfg.POS_FRACTION >= 0 else None
        self.sample_size = anchor_target_cfg.SAMPLE_SIZE
        self.norm_by_num_examples = anchor_target_cfg.NORM_BY_NUM_EXAMPLES
        self.matched_thresholds = {c['class_name']: c['matched_threshold'] for c in anchor_generator_cfg}
        self.unmatched_thresholds = {c['class_name']: c['unmatched_threshold'] for c in anchor_generator_cfg}
        self.use_multihead = model_cfg.get('USE_MULTIHEAD', False)
        if self.pos_fraction is not None or self.match_height or self.use_multihead:
            raise NotImplementedError("only the SECOND configuration (POS_FRACTION -1, MATCH_HEIGHT False, single "
                                      "head) is on the hot path")

    def _kernel_inputs(self, all_anchors):
        """Static device-side inputs of the fused HIP assigner (cached per device)."""
        dev = all_anchors[0].device
        cache = getattr(self, "_kcache", None)
        if cache is None or cache[0] != dev:
            shapes = {tuple(a.shape) for a in all_anchors}
            ok = len(shapes) == 1 and all_anchors[0].shape[-1] == 7 and self.box_coder.code_size == 7
            if ok:
                flat = torch.stack([a.reshape(-1, 7) for a in all_anchors], 0).contiguous().float()
                per_loc = all_anchors[0].shape[3] * all_anchors[0].shape[4]
                set_cls = torch.tensor([self.class_names.index(n) for n in self.anchor_class_names], dtype=torch.int32,
                                       device=dev)
                mt = torch.tensor([self.matched_thresholds[n] for n in self.anchor_class_names], dtype=torch.float32,
                                  device=dev)
                um = torch.tensor([self.unmatched_thresholds[n] for n in self.anchor_class_names], dtype=torch.float32,
                                  device=dev)
                cache = (dev, (flat, per_loc, set_cls, mt, um))
            else:
                cache = (dev, None)
            self._kcache = cache
        return cache[1]

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        """all_anchors: per class [1, H, W, n_size, n_rot, 7]; gt (B, M, 8).  Returns box_cls_labels [B, A] int32,
        box_reg_targets [B, A, code], reg_weights [B, A] with A ordered (y, x, class, size, rot).

        On the GPU this is libspx's fused assigner (csrc/assign.hip, two launches); the batched torch formulation
        below is the same rule and serves CPU tensors (tests) and anchor layouts the kernel does not cover."""
        gt = gt_boxes_with_classes
        if gt.is_cuda and gt.shape[-1] == 8 and gt.shape[1] <= 256:
            kin = self._kernel_inputs(all_anchors)
            if kin is not None:
                from spx import ops
                flat, per_loc, set_cls, mt, um = kin
                labels, targets, weights = ops.assign_targets(flat, per_loc, gt, set_cls, len(self.class_names), mt, um)
                return {'box_cls_labels': labels, 'box_reg_targets': targets, 'reg_weights': weights}
        return self.assign_targets_torch(all_anchors, gt)

    def assign_targets_torch(self, all_anchors, gt_boxes_with_classes):
        gt = gt_boxes_with_classes
        B, M = gt.shape[0], gt.shape[1]
        dev = gt.device
        gt_boxes = gt[..., :-1]
        gt_cls = gt[..., -1].int()
        # reference: drop trailing rows whose box part sums to 0, always keep row 0
        nz = gt_boxes.sum(-1) != 0                                               # [B, M]
        pos = torch.arange(M, device=dev).view(1, M)
        last = torch.where(nz, pos, torch.zeros_like(pos)).amax(dim=1, keepdim=True)
        kept = pos <= last                                                        # [B, M]
        n_cls = len(self.class_names)
        name_idx = torch.remainder(gt_cls.long() - 1, n_cls)                      # class id 0 -> class_names[-1]
        gt_bev = box_utils.boxes3d_lidar_to_aligned_bev_boxes(gt_boxes[..., 0:7])  # [B, M, 4]

        labels_all, targets_all, weights_all = [], [], []
        fmap = None
        for cname, anchors in zip(self.anchor_class_names, all_anchors):
            fmap = anchors.shape[:3]
            per_loc = anchors.shape[3] * anchors.shape[4]
            a = anchors.reshape(-1, anchors.shape[-1])                            # [A_c, 7]
            A = a.shape[0]
            valid = kept & (name_idx == self.class_names.index(cname))            # [B, M]
            iou = box_utils.boxes_iou_normal(box_utils.boxes3d_lidar_to_aligned_bev_boxes(a[:, 0:7]), gt_bev)
            iou = torch.where(valid[:, None, :], iou, iou.new_full((), -1.0))      # [B, A, M]
            a_max, a_arg = iou.max(dim=2)                                         # per anchor
            g_max = iou.amax(dim=1)                                               # per gt   [B, M]
            g_max = torch.where(g_max == 0, g_max.new_full((), -1.0), g_max)      # empty_gt_mask
            g_max = torch.where(valid, g_max, g_max.new_full((), -2.0))           # never equal to any IoU
            forced = (iou == g_max[:, None, :]).any(dim=2)                        # [B, A]
            has_gt = valid.any(dim=1, keepdim=True)                               # [B, 1]
            cls_of_arg = torch.gather(gt_cls, 1, a_arg)                           # [B, A]
            fg = (forced | (a_max >= self.matched_thresholds[cname])) & has_gt
            bg = (a_max < self.unmatched_thresholds[cname]) | ~has_gt
            labels = torch.full((B, A), -1, dtype=torch.int32, device=dev)
            labels = torch.where(bg, torch.zeros_like(labels), labels)
            labels = torch.where(fg, cls_of_arg, labels)
            pos_mask = labels > 0
            g_sel = torch.gather(gt_boxes, 1, a_arg[..., None].expand(B, A, gt_boxes.shape[-1]))
            enc = self.box_coder.encode_torch(g_sel, a[None].expand(B, A, a.shape[-1]))
            targets = torch.where(pos_mask[..., None], enc, torch.zeros_like(enc))
            weights = pos_mask.to(a.dtype)
            labels_all.append(labels.view(B, *fmap, per_loc))
            targets_all.append(targets.view(B, *fmap, per_loc, -1))
            weights_all.append(weights.view(B, *fmap, per_loc))
        code = targets_all[0].shape[-1]
        return {
            'box_cls_labels': torch.cat(labels_all, dim=-1).view(B, -1),
            'box_reg_targets': torch.cat(targets_all, dim=-2).view(B, -1, code),
            'reg_weights': torch.cat(weights_all, dim=-1).view(B, -1),
        }
