"""Generator / Discriminator / gradient_penalty with the reference's API
(architecture/gpsro/deeplab_gan.py) on the MI355X kernels."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import ops
from ...runtime import BGModule, pad_to, vec_of
from .deeplab import *  # noqa: F401,F403  (the reference re-exports the deeplab components the same way)
from .deeplab import Conv2d, DeepLabv3_plus, Xception


class Linear(BGModule):
    """Parameter container with nn.Linear's names for the single-logit head."""

    def __init__(self, in_features, out_features):
        super().__init__()
        assert out_features == 1
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features))


class Discriminator(BGModule):
    """Xception features -> flatten (NCHW order) -> Linear(F, 1) -> (logits, sigmoid)
    (deeplab_gan.py:12-39).  The reference hard-codes F = 12288 = 2048*2*3 (the
    19x37 grid); ``input_size=(H, W)`` sizes the head for any field."""

    def __init__(self, n_input=3, os=16, pretrained=False, normalizer=nn.LayerNorm, input_size=None,
                 compute_dtype=None):
        super().__init__()
        self.n_input = n_input
        self.xception_features = Xception(n_input, os, pretrained, normalizer)
        if input_size is None:
            feat = 12288
        else:
            h, w = int(input_size[0]), int(input_size[1])
            for _ in range(3 if os == 8 else 4):
                h, w = -(-h // 2), -(-w // 2)
            feat = 2048 * h * w
        self.linear = Linear(feat, 1)
        self.sigmoid = nn.Sigmoid()
        self._init_weight()
        if compute_dtype is not None:
            self.set_compute_dtype(compute_dtype)

    def _init_weight(self):
        # deeplab_gan.py:46-61: re-initialises every conv (Xception's included)
        gain = nn.init.calculate_gain("leaky_relu", 0.2)
        for m in self.modules():
            if isinstance(m, Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                nn.init.normal_(m.weight, mean=0.0, std=gain / math.sqrt(n))
            elif isinstance(m, Linear):
                nn.init.xavier_uniform_(m.weight, 1.0)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, input):
        dt = self.compute_dtype()
        if isinstance(input, (tuple, list)):
            # several batches pushed through as one (the trainer's D(real) + D(fake) pass): converted straight into the
            # batch slices of one internal buffer instead of torch.cat + one conversion
            n, c, H, W = sum(t.shape[0] for t in input), *input[0].shape[1:]
            assert c == self.n_input, f"expected {self.n_input} input channels, got {c}"
            xi = ops.ToInternalCat.apply(pad_to(c, vec_of(dt)), dt, *input)
        else:
            n, c, H, W = input.shape
            assert c == self.n_input, f"expected {self.n_input} input channels, got {c}"
            xi = ops.ToInternal.apply(input, pad_to(c, vec_of(dt)), dt)
        f, _ = self.xception_features.forward_nhwc(xi, want_low=False)
        a = self.arena()
        lin = self.linear
        if lin.in_features != f.shape[1] * f.shape[2] * 2048:
            raise RuntimeError(f"Discriminator head has in_features={lin.in_features} but the feature map is "
                               f"2048x{f.shape[1]}x{f.shape[2]}; construct it with input_size=({H}, {W})")
        logits = ops.LinearHeadFn.apply(f, lin.weight, lin.bias, a, a.by_param[id(lin.weight)], a.by_param[id(lin.bias)])
        return logits, torch.sigmoid(logits)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()


class Generator(BGModule):
    """noise concat + DeepLabv3_plus (deeplab_gan.py:64-94)."""

    def __init__(self, n_input, n_output, upsampler_type, noise_type, noise_dimensions, os=16, pretrained=False,
                 normalizer=nn.BatchNorm2d, compute_dtype=None):
        super().__init__()
        self.noise_dimensions = noise_dimensions
        if noise_type == "Uniform":
            self.dist = torch.distributions.uniform.Uniform(0., 1.)
        elif noise_type == "Normal":
            self.dist = torch.distributions.normal.Normal(0., 1.)
        else:
            raise NotImplementedError("Error, noise type {} not supported.".format(noise_type))
        self.noise_on_device = False  # True: draw with the device generator instead of the host RNG stream
        self.model = DeepLabv3_plus(n_input=(n_input + noise_dimensions), n_output=n_output, os=os,
                                    upsampler_type=upsampler_type, pretrained=pretrained, normalizer=normalizer)
        if compute_dtype is not None:
            self.set_compute_dtype(compute_dtype)

    def forward(self, input_raw, noise=None):
        if self.noise_dimensions > 0:
            shape = (input_raw.shape[0], self.noise_dimensions, input_raw.shape[2], input_raw.shape[3])
            if noise is None:
                if self.noise_on_device:
                    noise = (torch.rand if isinstance(self.dist, torch.distributions.Uniform) else torch.randn)(
                        shape, device=input_raw.device)
                else:  # the reference draws on the host RNG stream and copies (deeplab_gan.py:89)
                    noise = self.dist.rsample(shape).to(input_raw.device)
            input = torch.cat((input_raw, noise), dim=1)
        else:
            input = input_raw
        return self.model(input)


def gradient_penalty(critic, images_fake, images_real, eta=None):
    """WGAN-GP term exactly as the reference computes it (deeplab_gan.py:98-114):
    first-order only -- the result is a constant w.r.t. the critic's parameters."""
    if eta is None:
        eta = torch.distributions.uniform.Uniform(0., 1.).rsample((images_fake.shape[0], 1, 1, 1))
    eta = eta.to(images_fake.device)
    images_interpol = (eta * images_fake.detach() + (1. - eta) * images_real.detach()).requires_grad_(True)
    params = [p for p in critic.parameters() if p.requires_grad]
    for p in params:      # data-gradient only: skip every weight-gradient kernel
        p.requires_grad_(False)
    try:
        logits_interpol, _ = critic(images_interpol)
        gradients = torch.autograd.grad(outputs=logits_interpol, inputs=images_interpol,
                                        grad_outputs=torch.ones_like(logits_interpol))[0]
    finally:
        for p in params:
            p.requires_grad_(True)
    return ops.gp_penalty_value(gradients)
