"""GAN and pixel losses with the reference's API (utils/losses.py:101-172,
train_gan.py:142-152), evaluated by the HIP loss kernels."""
from __future__ import annotations

import torch

from .. import ops


class _PixelMean:
    def __init__(self, kind):
        self.kind = ops.PIXEL_LOSS_KINDS[kind]

    def __call__(self, prediction, target):
        return ops.L1LossFn.apply(prediction, target, None, 1.0 / prediction.numel(), self.kind)


def L1Loss():
    """Stand-in for nn.L1Loss() (mean absolute error), train_gan.py:146."""
    return _PixelMean("l1")


def SmoothL1Loss():
    """Stand-in for nn.SmoothL1Loss() (beta = 1), train_gan.py:148."""
    return _PixelMean("smooth_l1")


def MSELoss():
    """Stand-in for nn.MSELoss(), train_gan.py:150."""
    return _PixelMean("l2")


class _Weighted:
    kind = 0

    def __call__(self, prediction, target, weights):
        if self.normalize:
            s = ops.L1LossFn.apply(prediction, target, weights, 1.0, self.kind)
            return s / (torch.sum(weights) + self.eps)
        return ops.L1LossFn.apply(prediction, target, weights, 1.0 / prediction.numel(), self.kind)


class L1LossWeighted(_Weighted):
    """mean(|p-t|*w), or sum(|p-t|*w)/(sum(w)+eps) when normalize; smooth=True swaps in SmoothL1
    (losses.py:101-112)."""

    def __init__(self, normalize=False, eps=1.e-8, smooth=False):
        self.kind = ops.PIXEL_LOSS_KINDS["smooth_l1" if smooth else "l1"]
        self.eps = eps
        self.normalize = normalize


class L2LossWeighted(_Weighted):
    """mean((p-t)^2*w), or the weight-normalised sum (losses.py:115-126)."""

    def __init__(self, normalize=False, eps=1.e-8):
        self.kind = ops.PIXEL_LOSS_KINDS["l2"]
        self.eps = eps
        self.normalize = normalize


class GANLoss:
    """ModifiedMinMax (label-smoothed, 5 % label swap, BCE-with-logits) or Wasserstein
    (losses.py:129-172).  The three label draws of d_loss come from the host RNG in
    the reference's order (fake, real, swap), so a seeded run matches it draw for draw."""

    def __init__(self, mode, batch_size, device):
        self.mode = mode
        self.batch_size = batch_size
        self.device = device
        if self.mode == "ModifiedMinMax":
            self.label_real = torch.ones((self.batch_size, 1)).to(self.device)
            self.label_fake = torch.zeros((self.batch_size, 1)).to(self.device)
            self.dist_real = torch.distributions.uniform.Uniform(0.8, 1.0)
            self.dist_fake = torch.distributions.uniform.Uniform(0.0, 0.2)
            self.dist_swap = torch.distributions.uniform.Uniform(0.0, 1.0)
        elif self.mode == "Wasserstein":
            pass
        else:
            raise NotImplementedError("Error, {} loss not implemented".format(self.mode))

    def draw_labels(self):
        label_fake = self.dist_fake.rsample(self.label_fake.shape)
        label_real = self.dist_real.rsample(self.label_real.shape)
        swap = bool(self.dist_swap.sample() < 0.05)
        return label_fake, label_real, swap

    def d_loss(self, logits_real, logits_fake, labels=None):
        if self.mode == "ModifiedMinMax":
            label_fake, label_real, swap = labels if labels is not None else self.draw_labels()
            label_fake = label_fake.to(self.device, non_blocking=True)
            label_real = label_real.to(self.device, non_blocking=True)
            bce = ops.BCEWithLogitsFn.apply
            if swap:
                return 0.5 * (bce(logits_fake, label_real) + bce(logits_real, label_fake))
            return 0.5 * (bce(logits_fake, label_fake) + bce(logits_real, label_real))
        return torch.mean(logits_fake - logits_real)  # N scalars

    def g_loss(self, logits_fake):
        if self.mode == "ModifiedMinMax":
            return ops.BCEWithLogitsFn.apply(logits_fake, self.label_real)
        return -1. * torch.mean(logits_fake)
