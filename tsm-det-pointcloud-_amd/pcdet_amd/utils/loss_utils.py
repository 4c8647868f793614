"""Losses of the anchor head (semantics of reference pcdet/utils/loss_utils.py:9-77,140-209,310-338).
Device-agnostic: nothing is moved with .cuda() at construction."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class SigmoidFocalClassificationLoss(nn.Module):
    def __init__(self, gamma=2.0, alpha=0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        # max(x,0) - x*z + log(1 + exp(-|x|))
        return torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))

    def forward(self, input, target, weights):
        p = torch.sigmoid(input)
        alpha_w = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - p) + (1.0 - target) * p
        loss = alpha_w * torch.pow(pt, self.gamma) * self.sigmoid_cross_entropy_with_logits(input, target)
        if weights.dim() == 2 or (weights.dim() == 1 and target.dim() == 2):
            weights = weights.unsqueeze(-1)
        assert weights.dim() == loss.dim()
        return loss * weights


class WeightedSmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0 / 9.0, code_weights=None):
        super().__init__()
        self.beta = beta
        if code_weights is not None:
            self.register_buffer("code_weights", torch.from_numpy(np.array(code_weights, dtype=np.float32)),
                                 persistent=False)
        else:
            self.code_weights = None

    @staticmethod
    def smooth_l1_loss(diff, beta):
        if beta < 1e-5:
            return torch.abs(diff)
        n = torch.abs(diff)
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        target = torch.where(torch.isnan(target), input, target)
        diff = input - target
        if self.code_weights is not None:
            diff = diff * self.code_weights.to(diff.device).view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


class WeightedCrossEntropyLoss(nn.Module):
    def forward(self, input, target, weights):
        return F.cross_entropy(input.permute(0, 2, 1), target.argmax(dim=-1), reduction="none") * weights
